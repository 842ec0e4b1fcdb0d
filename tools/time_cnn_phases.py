"""Debug: per-phase cycle counts of cnn_bwd_wino_k (build with -DGNF_CNN_TIMING)."""
import ctypes, subprocess, sys, os
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, ROOT + '/graphical-normalizing-flows_amd']
src = ROOT + '/graphical-normalizing-flows_amd/gnf_hip/csrc/'
so = '/tmp/libgnf_timing.so'
subprocess.run(['hipcc', '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-shared', '-fno-slp-vectorize', '-DGNF_CNN_TIMING'] + sys.argv[1:] + ['-I' + ROOT + '/include', '-I' + src,
                src + 'gnf_mnistcnn_fwd.hip', src + 'gnf_mnistcnn.hip', src + 'gnf_rowwise.hip', '-o', so], check=True)
lib = ctypes.CDLL(so)
n = 78400
dev = 'cuda:0'
torch.manual_seed(0)
e = torch.randn(n, 784, device=dev) * (torch.rand(n, 784, device=dev) < .03).float()
W1, b1 = torch.randn(16, 9, device=dev) * .3, torch.randn(16, device=dev) * .1
W2, b2 = torch.randn(16, 144, device=dev) * .1, torch.randn(16, device=dev) * .1
pooled = torch.empty(n, 2304, device=dev); arg = torch.empty(n, 2304, dtype=torch.uint8, device=dev)
P = ctypes.c_void_p
st = P(torch.cuda.current_stream().cuda_stream)
lib.gnf_mnistcnn_conv_bwd_ws_bytes.restype = ctypes.c_int64
rc = lib.gnf_mnistcnn_conv_fwd(P(e.data_ptr()), P(W1.data_ptr()), P(b1.data_ptr()), P(W2.data_ptr()), P(b2.data_ptr()), P(pooled.data_ptr()), P(arg.data_ptr()), ctypes.c_int64(n), ctypes.c_int(0), st)
torch.cuda.synchronize()
tf = pooled[0, :64].view(8, 8).cpu()
fimgs = (n - 7 + 255) // 256
fnames = ["barrier wait", "stage + fetch (+ split finish)", "conv1 units (next image)", "whole item", "half item", "loop"]
print("forward (cnn_fwd_wino_k), images per WG", fimgs)
for w in range(8):
    print("wave", w, {fnames[k]: int(tf[w, k].item() / fimgs) for k in range(6)}, "total/img", int(tf[w, :6].sum().item() / fimgs))
rc = lib.gnf_mnistcnn_conv_fwd(P(e.data_ptr()), P(W1.data_ptr()), P(b1.data_ptr()), P(W2.data_ptr()), P(b2.data_ptr()), P(pooled.data_ptr()), P(arg.data_ptr()), ctypes.c_int64(n), ctypes.c_int(0), st)
nws = lib.gnf_mnistcnn_conv_bwd_ws_bytes(ctypes.c_int64(n))
ws = torch.zeros(nws // 4, device=dev)
gp = torch.randn(n, 2304, device=dev); ge = torch.empty(n, 784, device=dev)
g = [torch.empty_like(t) for t in (W1, b1, W2, b2)]
for _ in range(2):
    rc = lib.gnf_mnistcnn_conv_bwd(P(e.data_ptr()), P(W1.data_ptr()), P(b1.data_ptr()), P(W2.data_ptr()), P(gp.data_ptr()), P(arg.data_ptr()), P(ge.data_ptr()),
                                   P(g[0].data_ptr()), P(g[1].data_ptr()), P(g[2].data_ptr()), P(g[3].data_ptr()), P(ws.data_ptr()), ctypes.c_int64(nws), ctypes.c_int64(n), st)
torch.cuda.synchronize()
PROW = 16 * 144 + 256 + 16
t = ws[(256 * 8 + 1) * PROW:(256 * 8 + 1) * PROW + 64].view(8, 8).cpu()
names = ["B3 wait", "stage + prefetch", "dW2", "de(prev) + B2 wait", "conv1 first", "da1 group 0", "da1 group 1", "conv1 last + scatter(next)"]
imgs = (n - 7 + 255) // 256
print("rc", rc, "images per WG", imgs)
for w in range(8):
    print("wave", w, {names[k]: int(t[w, k].item() / imgs) for k in range(8)}, "total/img", int(t[w, :8].sum().item() / imgs))

