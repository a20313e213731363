# shader clock while bench.py runs in another process: tools/clock_probe samples beside it
mkdir -p gpurun_out/r04f
tools/clock_probe 40 > gpurun_out/r04f/clock_load.txt 2>&1 &
CP=$!
sleep 3
python bench.py --steps 200 --repeats 150 --no-cpu-baseline --no-e2e > gpurun_out/r04f/clock_load_bench.json 2> gpurun_out/r04f/clock_load_bench.err
python -c "import json; d=json.load(open('gpurun_out/r04f/clock_load_bench.json')); print('bench beside the sampler:', d['value'], 'frames/s')" >> gpurun_out/r04f/clock_load.txt
wait $CP
cat gpurun_out/r04f/clock_load.txt
