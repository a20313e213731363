#!/bin/bash
# One GPU session that produces every file profiles/ cites for a round (run from the repo root on the GPU box):
#   gpurun --timeout 1100 -- 'bash tools/profile_round.sh r02'
# rocprofv3 is given the program itself after `--` (python3 bench.py ...), PMC passes are separate runs without any trace
# domain but --kernel-trace (gpurun refuses other combinations); FETCH_SIZE and WRITE_SIZE need a pass each (TCC slots).
R=${1:-r05}
O=gpurun_out/$R
mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
B="--no-cpu-baseline --no-e2e --no-boundary-legs"
# 1. bench lines (the default command first: it is what the driver runs)
python3 bench.py > $O/bench_default.json 2> $O/bench_default.err
python3 bench.py --streams 1 $B > $O/bench_1stream.json 2>/dev/null
python3 bench.py --no-fuse $B > $O/bench_nofuse.json 2>/dev/null
for w in nrdm_3_540p sesr_x4_540p sesr_x4_540p_b32 nrdm6_sesrx2_540p; do python3 bench.py --workload $w --steps 100 > $O/bench_$w.json 2>/dev/null; done
python3 bench.py --steps 20 --warmup 5 > $O/bench_driver_form.json 2>/dev/null      # the driver's command line
# 2. kernel traces with statistics: default (2 streams, overlapped) and 1 stream (kernels alone)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_default -- python3 bench.py --steps 50 --warmup 5 --repeats 2 $B > $O/trace_default.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/trace_1stream -- python3 bench.py --steps 50 --warmup 5 --repeats 2 --streams 1 $B > $O/trace_1stream.log 2>&1
cp $(find $O/trace_default -name "*kernel_stats.csv" | head -1) $O/${R}_default_kernel_stats.csv
cp $(find $O/trace_1stream -name "*kernel_stats.csv" | head -1) $O/${R}_1stream_kernel_stats.csv
# 3. PMC passes (1 stream: per-launch values of kernels that run alone)
pass() { n=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $O/pmc_$n -- python3 bench.py --steps 5 --warmup 2 --repeats 1 --streams 1 --submit step $B > $O/pmc_$n.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass sq1 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU SQ_INSTS_MFMA
pass sq2 SQ_VALU_MFMA_BUSY_CYCLES SQ_INSTS_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_WAVES
pass grbm GRBM_GUI_ACTIVE
python3 tools/pmc_summary.py $O > $O/pmc_summary.log 2>&1
tail -20 $O/pmc_summary.log
