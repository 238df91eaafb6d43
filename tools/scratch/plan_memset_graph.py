import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import _lib
lib = _lib.load()
rows, C = 128, 256
def build(twod):
    plan = ctypes.c_void_p(); lib.sn_plan_create(ctypes.byref(plan))
    if twod:   # zero the first half of a (rows, C) buffer: pitch C*4, width C/2*4
        assert lib.sn_plan_add_memset(plan, 0, 0, 0, C * 4, C // 2 * 4, rows) == 0
    else:
        assert lib.sn_plan_add_memset(plan, 0, 0, 0, rows * C * 4, rows * C * 4, 1) == 0
    n = 7; I32, I64, F64 = ctypes.c_int32 * n, ctypes.c_int64 * n, ctypes.c_double * n
    fn = lib.sn_plan_lookup(b"sn_elu_into_f32")
    assert lib.sn_plan_add_call(plan, fn, n, I32(2, 0, 2, 0, 0, 0, 4), I32(0, 0, 1, 0, 0, 0, 0), I64(0, C, 0, C, rows, C, 0), F64()) == 0
    return plan
for twod in (False, True):
    plan = build(twod)
    x = torch.full((rows, C), 3.0, device="cuda"); y = torch.empty((rows, C), device="cuda")
    bases = (ctypes.c_uint64 * 2)(x.data_ptr(), y.data_ptr())
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        assert lib.sn_plan_run(plan, bases, 2, s.cuda_stream, None) == 0
    torch.cuda.synchronize()
    print("eager twod", twod, "zero part max", float(y[:, :C // 2].abs().max()), "rest", float(y[:, C // 2:].min()))
    g = torch.cuda.CUDAGraph()
    x.fill_(3.0)
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        assert lib.sn_plan_run(plan, bases, 2, st, None) == 0
    for rep in range(3):
        x.fill_(5.0 + rep); y.fill_(-1.0)
        g.replay(); torch.cuda.synchronize()
        print("  replay", rep, "zero part max", float(y[:, :C // 2].abs().max()), "rest", float(y[:, C // 2:].min()), float(y[:, C // 2:].max()))

# which flat hipMemsetAsync sizes replay wrongly from a captured graph?
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemsetAsync.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t, ctypes.c_void_p]
for n in (4, 12, 256, 1024, 4096, 65536, 131072, 1 << 20, 1 << 24, 1 << 28):
    buf = torch.empty(n, dtype=torch.uint8, device="cuda")
    out = torch.empty(n, dtype=torch.uint8, device="cuda")
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        st = torch.cuda.current_stream().cuda_stream
        assert hip.hipMemsetAsync(buf.data_ptr(), 0, n, st) == 0
        out.copy_(buf)
    bad = []
    for rep in range(4):
        buf.fill_(7 + rep); out.fill_(1)
        g.replay(); torch.cuda.synchronize()
        bad.append(int((out != 0).sum()))
    print("hipMemsetAsync bytes", n, "nonzero after replays", bad)
