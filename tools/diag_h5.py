import sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), 'tests')); sys.path.insert(0, os.path.join(os.getcwd(), 'sesr-pytorch-quantize_amd'))
from helpers import bundle_from_oracle, fixture_case, rand_frame
from oracle import sesrq_oracle as O, c_oracle as CO
import sesrq
fx, meta, net, _ = fixture_case('tests/golden/sesr_x2_rand.crop.npz')
for (H, W, budget) in ((270, 480, 0), (540, 960, 0), (1080, 1920, 512), (1080, 1920, 0)):
    x = rand_frame((1, 3, H, W), 5)
    e = sesrq.Engine(bundle_from_oracle(net), torch.device('cuda:0'), wg_budget=budget)
    q, y = e.forward(torch.from_numpy(x).cuda())
    want = CO.forward(net, x, want_f=False)['q_out']
    got = q.cpu().numpy()
    N,C,Ho,Wo = got.shape
    def unsh(a): return a.reshape(N,C,Ho//2,2,Wo//2,2).transpose(0,1,3,5,2,4).reshape(N,C*4,Ho//2,Wo//2)
    b = (unsh(got) != unsh(want))
    print(H, W, budget, 'mismatches', int(b.sum()), 'of', got.size)
    if b.sum():
        print(' per channel', b.sum(axis=(0,2,3)))
        r = b.sum(axis=(0,1,3)); c = b.sum(axis=(0,1,2))
        print(' rows with errors', np.nonzero(r)[0][:40], len(np.nonzero(r)[0]))
        print(' cols with errors', np.nonzero(c)[0][:80], len(np.nonzero(c)[0]))
        print(' per row mod 8', [int(r[k::8].sum()) for k in range(8)]); print(' per col mod 64', [int(c[k::64].sum()) for k in range(64)])
