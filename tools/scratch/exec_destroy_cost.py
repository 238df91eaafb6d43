import ctypes, os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from surfacenetworks_amd import _lib
lib = _lib.load()
rows, C = 257, 128
plan = ctypes.c_void_p(); lib.sn_plan_create(ctypes.byref(plan))
n = 7; I32, I64, F64 = ctypes.c_int32 * n, ctypes.c_int64 * n, ctypes.c_double * n
for _ in range(8):
    lib.sn_plan_add_call(plan, lib.sn_plan_lookup(b"sn_elu_into_f32"), n, I32(2, 0, 2, 0, 0, 0, 4), I32(0, 0, 1, 0, 0, 0, 0), I64(0, C, 0, C, rows, C, 0), F64())
x = torch.randn(rows, C, device="cuda"); ys = [torch.empty(rows, C, device="cuda") for _ in range(60)]
execs = []
t0 = time.perf_counter()
for y in ys:
    b = (ctypes.c_uint64 * 2)(x.data_ptr(), y.data_ptr()); ex = ctypes.c_void_p()
    assert lib.sn_plan_instantiate(plan, b, 2, ctypes.byref(ex), None) == 0
    execs.append((ex, b))
t1 = time.perf_counter()
s = torch.cuda.current_stream().cuda_stream
for ex, b in execs: lib.sn_plan_exec_launch(ex, plan, b, 2, s, None)
torch.cuda.synchronize()
t2 = time.perf_counter()
for ex, b in execs: lib.sn_plan_exec_destroy(ex)
t3 = time.perf_counter()
print(f"instantiate {1e6*(t1-t0)/60:.0f} us each, first launch {1e6*(t2-t1)/60:.0f} us each (incl. sync), destroy {1e6*(t3-t2)/60:.0f} us each")
