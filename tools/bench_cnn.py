"""A/B timing of the dense conv kernels: builds gnf_mnistcnn.hip (+ gnf_rowwise.hip) with the given extra hipcc flags
into a scratch library and times the forward / backward entry points at the cfg4 size (78 400 images) with HIP events.
    python tools/bench_cnn.py [-DFOO ...] [--label NAME]"""
import ctypes, subprocess, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
src = ROOT + '/graphical-normalizing-flows_amd/gnf_hip/csrc/'
flags = [a for a in sys.argv[1:] if a.startswith('-') and a != '--label']
label = sys.argv[sys.argv.index('--label') + 1] if '--label' in sys.argv else ' '.join(flags) or 'default'
so = '/tmp/libgnf_cnn_ab_%d.so' % os.getpid()
sys.path.insert(0, ROOT + '/graphical-normalizing-flows_amd')
from gnf_hip.build import EXTRA_FLAGS   # the product's per-file flags apply here too
from _warm import warm_gpu  # noqa: E402
objs = []
for f in (os.environ.get('GNF_CNN_FWD_SRC', 'gnf_mnistcnn_fwd.hip'), os.environ.get('GNF_CNN_BWD_SRC', 'gnf_mnistcnn.hip'),
          'gnf_rowwise.hip'):                                       # A/B against other sources in csrc/
    o = '/tmp/cnn_ab_%d_%s.o' % (os.getpid(), f)
    subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-Wno-unused-value'] + EXTRA_FLAGS.get(f, EXTRA_FLAGS.get('gnf_mnistcnn_fwd.hip' if 'fwd' in f else 'gnf_mnistcnn.hip', []) if 'mnistcnn' in f else []) + flags +
                   ['-I' + ROOT + '/include', '-I' + src, '-c', src + f, '-o', o], check=True)
    objs.append(o)
subprocess.run(['hipcc', '--offload-arch=gfx950', '-shared', '-fPIC', '-o', so] + objs, check=True)
for o in objs:
    os.remove(o)
lib = ctypes.CDLL(so)
n = 78400
dev = 'cuda:0'
torch.manual_seed(0)
e = torch.randn(n, 784, device=dev) * .5
W1, b1 = torch.randn(16, 9, device=dev) * .3, torch.randn(16, device=dev) * .1
W2, b2 = torch.randn(16, 144, device=dev) * .1, torch.randn(16, device=dev) * .1
pooled = torch.empty(n, 2304, device=dev); arg = torch.empty(n, 2304, dtype=torch.uint8, device=dev)
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)
lib.gnf_mnistcnn_conv_bwd_ws_bytes.restype = ctypes.c_int64
nws = lib.gnf_mnistcnn_conv_bwd_ws_bytes(ctypes.c_int64(n))
ws = torch.zeros(nws // 4, device=dev)
gp = torch.randn(n, 2304, device=dev); ge = torch.empty(n, 784, device=dev)
g = [torch.empty_like(t) for t in (W1, b1, W2, b2)]


def fwd():
    return lib.gnf_mnistcnn_conv_fwd(P(e.data_ptr()), P(W1.data_ptr()), P(b1.data_ptr()), P(W2.data_ptr()), P(b2.data_ptr()),
                                     P(pooled.data_ptr()), P(arg.data_ptr()), ctypes.c_int64(n), ctypes.c_int(0), st)


def bwd():
    return lib.gnf_mnistcnn_conv_bwd(P(e.data_ptr()), P(W1.data_ptr()), P(b1.data_ptr()), P(W2.data_ptr()), P(gp.data_ptr()),
                                     P(arg.data_ptr()), P(ge.data_ptr()), P(g[0].data_ptr()), P(g[1].data_ptr()),
                                     P(g[2].data_ptr()), P(g[3].data_ptr()), P(ws.data_ptr()), ctypes.c_int64(nws),
                                     ctypes.c_int64(n), st)


# round 5: the compact-de variant (gnf_mnistcnn_conv_bwd_cols) with the plan of MNIST_A_prior(28, 2), built on the host here
# (the product builds it on the device in gnf_dag_gate_fwd_plan)
KC = 32
r_, c_ = torch.arange(28).repeat_interleave(28), torch.arange(28).repeat(28)
win = ((r_[:, None] - r_[None, :]).abs() <= 2) & ((c_[:, None] - c_[None, :]).abs() <= 2)
win.fill_diagonal_(False)
cols = torch.full((784, KC), -1, dtype=torch.int16)
for i in range(784):
    jj = win[i].nonzero().flatten()
    cols[i, :len(jj)] = jj.to(torch.int16)
plan = torch.cat([win.sum(1).to(torch.int32), cols.view(-1).view(torch.int32), torch.zeros(1, dtype=torch.int32)]).to(dev)  # counts | cols | overflow word
gec = torch.empty(n, KC, device=dev)
HAS_COLS = hasattr(lib, 'gnf_mnistcnn_conv_bwd_cols')


def bwd_cols():
    return lib.gnf_mnistcnn_conv_bwd_cols(P(e.data_ptr()), P(W1.data_ptr()), P(b1.data_ptr()), P(W2.data_ptr()), P(gp.data_ptr()),
                                          P(arg.data_ptr()), P(ge.data_ptr()), P(plan.data_ptr()), ctypes.c_int64(784),
                                          P(gec.data_ptr()), P(g[0].data_ptr()), P(g[1].data_ptr()),
                                          P(g[2].data_ptr()), P(g[3].data_ptr()), P(ws.data_ptr()), ctypes.c_int64(nws),
                                          ctypes.c_int64(n), st)


def timeit(fn, reps=15):
    warm_gpu()
    for _ in range(3):
        assert fn() == 0
    ts = []
    for _ in range(reps):
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        ts.append(a.elapsed_time(b))
    ts.sort()
    return ts[len(ts) // 2]


tf, tb = timeit(fwd), timeit(bwd)
chk = (pooled.double().sum().item(), ge.double().sum().item(), g[2].double().sum().item(), g[0].double().sum().item())
print("[%s] conv fwd %.4f ms   conv bwd %.4f ms   checksums pooled %.6e ge %.6e gW2 %.6e gW1 %.6e" % ((label, tf, tb) + chk))
if HAS_COLS:
    g2d, g0d = g[2].clone(), g[0].clone()
    tc = timeit(bwd_cols)
    rows = torch.arange(n, device=dev) % 784
    cd = cols.to(dev).long()
    bad = 0
    for k in range(KC):
        on = cd[rows, k] >= 0
        bad += int((gec[on, k] != ge[on.nonzero().flatten(), cd[rows, k][on]]).sum())
    print("[%s] conv bwd, compact de (MNIST window plan) %.4f ms   mismatching ge_cols entries %d   gW2 equal %s gW1 equal %s"
          % (label, tc, bad, bool(torch.equal(g2d, g[2])), bool(torch.equal(g0d, g[0]))))
os.remove(so)
