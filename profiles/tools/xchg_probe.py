import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29544")
import numpy as np, torch, torch.distributed as dist, ctypes as C
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
from flashweave_jl_amd.dist import make_allgather
st = {}
cb = make_allgather(dist, torch.device("cuda", 0), stats=st)
for n in (9000, 9000, 9000, 20000, 9000):
    t = np.arange(n, dtype=np.int32); u = np.arange(n, dtype=np.int32) + 5
    s = np.random.rand(n); p = np.random.rand(n)
    nt = C.c_int64(0); pt, pn = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)(); ps, pp = C.POINTER(C.c_double)(), C.POINTER(C.c_double)()
    for rep in range(5):
        t0 = time.perf_counter()
        cb(None, n, t.ctypes.data_as(C.POINTER(C.c_int32)), u.ctypes.data_as(C.POINTER(C.c_int32)), s.ctypes.data_as(C.POINTER(C.c_double)), p.ctypes.data_as(C.POINTER(C.c_double)), C.byref(nt), C.byref(pt), C.byref(pn), C.byref(ps), C.byref(pp))
        dt = time.perf_counter() - t0
    print(n, "last call us", round(dt * 1e6, 1))
dist.destroy_process_group()
