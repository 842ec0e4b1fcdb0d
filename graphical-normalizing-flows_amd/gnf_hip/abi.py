"""ctypes binding of include/gnf_hip.h (libgnf_hip.so).

PyTorch is used only as the owner of device memory and streams: every entry point
receives raw `data_ptr()`s, element strides and torch's current HIP stream.  There is
NO CPU fallback: a missing library or a non-HIP tensor raises."""
import ctypes
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libgnf_hip.so")

c_f = ctypes.c_void_p            # device float*
c_i64 = ctypes.c_int64
c_int = ctypes.c_int
c_float = ctypes.c_float
c_double = ctypes.c_double
c_u64 = ctypes.c_uint64
c_stream = ctypes.c_void_p

MONO_MAX_LAYERS = 8
DAG_PLAN_KC = 32          # GNF_DAG_PLAN_KC
ABI_VERSION = 8           # GNF_ABI_VERSION of include/gnf_hip.h this binding was written against


class MonoNet(ctypes.Structure):
    """gnf_mono_net (include/gnf_hip.h)."""
    _fields_ = [("nl", c_int), ("dims", c_int * (MONO_MAX_LAYERS + 1)),
                ("W", ctypes.c_void_p * MONO_MAX_LAYERS), ("b", ctypes.c_void_p * MONO_MAX_LAYERS)]


# name -> (restype, argtypes); must list every symbol include/gnf_hip.h declares
SIGNATURES = {
    "gnf_abi_version": (c_int, []),
    "gnf_affine_fwd": (c_int, [c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_f, c_f, c_int, c_i64, c_i64, c_stream]),
    "gnf_affine_bwd": (c_int, [c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_i64, c_i64,
                               c_i64, c_stream]),
    "gnf_affine_inv": (c_int, [c_f, c_f, c_i64, c_i64, c_i64, c_f, c_i64, c_i64, c_stream]),
    "gnf_logsum_rows_fwd": (c_int, [c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_logsum_rows_bwd": (c_int, [c_f, c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_normal_logdensity_fwd": (c_int, [c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_normal_logdensity_bwd": (c_int, [c_f, c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_nll_reduce_fwd": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_nll_reduce_bwd": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_nll_mean_fwd": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_stream]),
    "gnf_nll_mean_bwd": (c_int, [c_f, c_f, c_f, c_i64, c_stream]),
    "gnf_nll_loss_max_elems": (c_i64, []),
    "gnf_nll_loss_fwd": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_nll_loss_bwd": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_colsum_ws_bytes": (c_i64, [c_i64, c_i64]),
    "gnf_colsum": (c_int, [c_f, c_i64, c_f, c_i64, c_i64, c_f, c_stream]),
    "gnf_linear_ws_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "gnf_linear_fwd": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_int, c_f, c_i64, c_i64, c_i64, c_f, c_i64, c_stream]),
    "gnf_linear_bwd_x": (c_int, [c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_f, c_i64, c_i64, c_i64, c_f, c_i64, c_stream]),
    "gnf_linear_bwd_w": (c_int, [c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_f, c_i64, c_i64, c_i64, c_f, c_i64, c_stream]),
    "gnf_linear_bwd": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, c_int, c_f, c_f, c_f, c_f, c_f, c_i64, c_i64, c_i64, c_f, c_i64,
                               c_stream]),
    "gnf_linear_gxsum_fused": (c_int, [c_i64, c_i64, c_i64, c_int]),
    "gnf_gemm_last_kernel": (ctypes.c_char_p, []),
    "gnf_gemm_ws_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "gnf_gemm_f32_ws_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "gnf_gemm": (c_int, [c_f, c_i64, c_i64, c_f, c_f, c_i64, c_i64, c_f, c_i64, c_i64, c_f, c_f, c_i64, c_i64, c_f,
                         c_i64, c_i64, c_int, c_i64, c_i64, c_i64, c_f, c_i64, c_stream]),
    "gnf_gemm_split_ws_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "gnf_gemm_split_last_kernel": (ctypes.c_char_p, []),
    "gnf_gemm_split_enabled": (c_int, []),
    "gnf_gemm_split_bf16": (c_int, [c_f, c_i64, c_i64, c_f, c_i64, c_i64, c_f, c_i64, c_i64, c_f, c_int, c_i64, c_i64, c_i64,
                                    c_int, c_int, c_i64, ctypes.c_void_p, c_i64, c_stream]),
    "gnf_dag_gate_fwd_ws_bytes": (c_i64, [c_i64]),
    "gnf_dag_gate_fwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_float, c_float, c_f, c_f, c_u64, c_u64, c_int,
                                 c_f, c_i64, c_i64, c_stream]),
    "gnf_dag_gate_bwd_ws_bytes": (c_i64, [c_i64, c_i64]),
    "gnf_dag_gate_bwd": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_float, c_float, c_f, c_f, c_u64, c_u64, c_f, c_f,
                                 c_f, c_f, c_i64, c_i64, c_stream]),
    "gnf_dag_gate_plan_bytes": (c_i64, [c_i64]),
    "gnf_dag_gate_fwd_plan": (c_int, [c_f, c_f, c_f, c_i64, c_int, c_int, c_float, c_float, c_f, c_f, c_u64, c_u64, c_int,
                                      c_f, ctypes.c_void_p, c_i64, c_i64, c_i64, c_stream]),
    "gnf_dag_gate_bwd_cols_ws_bytes": (c_i64, [c_i64, c_i64]),
    "gnf_dag_gate_bwd_cols": (c_int, [c_f, c_f, c_f, ctypes.c_void_p, c_int, c_int, c_float, c_f, c_f, c_u64, c_u64, c_f,
                                      c_f, c_int, c_f, c_i64, c_i64, c_stream]),
    "gnf_monotonic_pack_floats": (c_i64, [ctypes.POINTER(MonoNet)]),
    "gnf_monotonic_pack": (c_int, [ctypes.POINTER(MonoNet), c_f, c_stream]),
    "gnf_monotonic_fwd": (c_int, [c_f, ctypes.POINTER(MonoNet), c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_int, c_f,
                                  c_f, c_i64, c_i64, c_stream]),
    "gnf_monotonic_fwd_f32": (c_int, [c_f, ctypes.POINTER(MonoNet), c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_int, c_f,
                                      c_f, c_i64, c_i64, c_stream]),
    "gnf_monotonic_fwd_kernel": (ctypes.c_char_p, []),
    "gnf_monotonic_inv": (c_int, [c_f, ctypes.POINTER(MonoNet), c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_int, c_f,
                                  c_i64, c_i64, c_stream]),
    "gnf_monotonic_inv_scatter": (c_int, [c_f, ctypes.POINTER(MonoNet), c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_int, c_f,
                                          ctypes.c_void_p, c_i64, c_i64, c_i64, c_stream]),
    "gnf_monotonic_bwd_ws_bytes": (c_i64, [ctypes.POINTER(MonoNet), c_int, c_i64, c_i64]),
    "gnf_monotonic_bwd": (c_int, [c_f, ctypes.POINTER(MonoNet), c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_int, c_f,
                                  c_f, c_f, c_f, c_i64, c_i64, c_i64, ctypes.POINTER(ctypes.c_void_p),
                                  ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, c_i64, c_i64, c_i64, c_stream]),
    "gnf_monotonic_bwd_f32": (c_int, [c_f, ctypes.POINTER(MonoNet), c_f, c_f, c_i64, c_i64, c_i64, c_f, c_f, c_int, c_f,
                                      c_f, c_f, c_f, c_i64, c_i64, c_i64, ctypes.POINTER(ctypes.c_void_p),
                                      ctypes.POINTER(ctypes.c_void_p), ctypes.c_void_p, c_i64, c_i64, c_i64, c_stream]),
    "gnf_monotonic_bwd_kernel": (ctypes.c_char_p, []),
    "gnf_dag_loss_prep": (c_int, [c_f, c_f, c_float, c_f, c_i64, c_stream]),
    "gnf_dag_loss_value": (c_int, [c_f, c_f, c_f, c_f, c_float, c_f, c_f, c_f, c_f, c_int, c_f, c_f, c_i64, c_stream]),
    "gnf_dag_loss_bwd": (c_int, [c_f, c_f, c_f, c_f, c_f, c_i64, c_stream]),
    "gnf_mnistcnn_conv_fwd": (c_int, [c_f, c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p, c_i64, c_int, c_stream]),
    "gnf_mnistcnn_conv_bwd_ws_bytes": (c_i64, [c_i64]),
    "gnf_mnistcnn_conv_bwd": (c_int, [c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p, c_f, c_f, c_f, c_f, c_f,
                                      ctypes.c_void_p, c_i64, c_i64, c_stream]),
    "gnf_mnistcnn_conv_bwd_cols": (c_int, [c_f, c_f, c_f, c_f, c_f, ctypes.c_void_p, c_f, ctypes.c_void_p, c_i64, c_f,
                                           c_f, c_f, c_f, c_f, ctypes.c_void_p, c_i64, c_i64, c_stream]),
    "gnf_mnistcnn_sparse_ws_bytes": (c_i64, [c_i64, c_i64]),
    "gnf_mnistcnn_sparse_fwd": (c_int, [c_f, c_i64, c_f, ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64, c_f, c_f, c_f,
                                        c_f, c_f, c_f, c_i64, c_f, c_f, ctypes.c_void_p, ctypes.c_void_p, c_i64,
                                        c_stream]),
    "gnf_mnistcnn_sparse_fwd_train": (c_int, [c_f, c_i64, c_f, ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64, c_f, c_f, c_f,
                                              c_f, c_f, c_f, c_i64, c_f, c_f, ctypes.c_void_p, ctypes.c_void_p, c_i64,
                                              c_stream]),
    "gnf_mnistcnn_sparse_prep_bytes": (c_i64, [c_i64]),
    "gnf_mnistcnn_sparse_prepare": (c_int, [c_f, c_f, c_f, c_f, c_f, c_i64, ctypes.c_void_p, c_i64, c_stream]),
    "gnf_mnistcnn_sparse_fwd_prepared": (c_int, [c_f, c_i64, c_f, ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64, c_f, c_f,
                                                 c_f, c_f, c_i64, ctypes.c_void_p, c_f, ctypes.c_void_p, c_i64, c_stream]),
    "gnf_mnistcnn_sparse_fwd_prepared_fc2": (c_int, [c_f, c_i64, c_f, ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64, c_f, c_f,
                                                     c_f, c_f, c_i64, ctypes.c_void_p, c_f, c_f, c_i64, c_f, ctypes.c_void_p,
                                                     c_i64, c_stream]),
    "gnf_mnistcnn_sparse_bwd_ws_bytes": (c_i64, [c_i64, c_i64, c_i64]),
    "gnf_mnistcnn_sparse_bwd": (c_int, [c_f, c_i64, c_f, ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64,
                                        ctypes.c_void_p, c_i64, ctypes.c_void_p, c_f, c_f, c_f,
                                        c_f, c_f, c_i64, c_f, ctypes.c_void_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f,
                                        ctypes.c_void_p, c_i64, c_stream]),
    "gnf_mnistcnn_sparse_bwd_tables": (c_int, [c_f, c_i64, c_f, ctypes.c_void_p, c_i64, ctypes.c_void_p, c_i64,
                                               ctypes.c_void_p, c_i64, ctypes.c_void_p, c_f, c_f, c_f,
                                               c_f, c_f, c_i64, ctypes.c_void_p, c_f, ctypes.c_void_p, c_f, c_f, c_f, c_f, c_f, c_f, c_f,
                                               ctypes.c_void_p, c_i64, c_stream]),
    "gnf_adam_step": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_double, c_double, c_double, c_double, c_double, c_double, c_int,
                              c_stream]),
    "gnf_adam_step_dev": (c_int, [c_f, c_f, c_f, c_f, c_i64, c_double, c_double, c_double, c_double, c_double, c_double,
                                  ctypes.c_void_p, c_int, c_stream]),
    "gnf_probe_mfma_f32": (c_i64, [c_f, c_int, c_int, c_stream]),
    "gnf_probe_copy": (c_int, [c_f, c_f, c_i64, c_stream]),
    "gnf_probe_empty": (c_int, [c_i64, c_int, c_stream]),
}

_lib = None


class GnfError(RuntimeError):
    pass


def load():
    """dlopen libgnf_hip.so (after `import torch`, so that the HIP runtime torch already
    loaded -- same SONAME libamdhip64.so.7 -- is the one the kernels launch on)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError("libgnf_hip.so is not built: run `python __graft_entry__.py build` "
                              "(hipcc --offload-arch=gfx950).  There is no CPU fallback. [%s]" % LIB_PATH)
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)       # AttributeError if the symbol is missing
            fn.restype = res
            fn.argtypes = args
        if lib.gnf_abi_version() != ABI_VERSION:
            raise ImportError("libgnf_hip.so ABI version mismatch")
        _lib = lib
    return _lib


_ERR = {-1: "GNF_EINVAL (bad argument)", -2: "GNF_ESHAPE (unsupported shape)", -3: "GNF_EWS (workspace too small)"}


def check(rc, what):
    if rc != 0:
        raise GnfError("%s failed: %s" % (what, _ERR.get(rc, "hipError_t %d" % rc)))


def stream():
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def ptr(t):
    """device pointer of a fp32 HIP tensor (None -> NULL)."""
    if t is None:
        return None
    if not t.is_cuda:
        raise GnfError("gnf_hip kernels run on the MI355X only: got a %s tensor (no CPU fallback)" % t.device)
    if t.dtype != torch.float32:
        raise GnfError("gnf_hip kernels are fp32: got %s" % t.dtype)
    return ctypes.c_void_p(t.data_ptr())


def rawptr(t):
    """device pointer of a HIP tensor of any dtype (byte buffers)."""
    if not t.is_cuda:
        raise GnfError("gnf_hip kernels run on the MI355X only: got a %s tensor (no CPU fallback)" % t.device)
    return ctypes.c_void_p(t.data_ptr())


_prof = None        # name -> [(start_event, end_event), ...] while profiling is enabled


def profile_enable(names):
    """Record HIP events (on torch's current stream = the stream the kernels are launched
    on) around every call of the named entry points; bench.py derives roofline numbers."""
    global _prof
    _prof = {n: [] for n in names}


def profile_collect():
    """-> {name: mean milliseconds per call}; disables profiling.  Synchronises."""
    global _prof
    out = {}
    if _prof:
        torch.cuda.synchronize()
        for n, evs in _prof.items():
            if evs:
                out[n] = sum(a.elapsed_time(b) for a, b in evs) / len(evs)
    _prof = None
    return out


def call(name, *args):
    fn = getattr(load(), name)
    if _prof is not None and name in _prof:
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        rc = fn(*args)
        b.record()
        _prof[name].append((a, b))
        check(rc, name)
    else:
        check(fn(*args), name)
