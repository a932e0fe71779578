import sys, os, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo")); sys.path.insert(0, os.path.join(sys.path[0], "tests"))
from conftest import build_model, load_golden, rmse
import test_gpu_model as T
prec = sys.argv[1]
gd = load_golden("traj_mc320_v4_d1_50steps_f64"); g32 = load_golden("traj_mc320_v4_d1_50steps")
m = build_model(320, precision=prec)
import time
_, _, _, x, inter = T._sample(m, 4, 1, 11, 50)
torch.cuda.synchronize(); t0=time.time()
_, _, _, x, inter = T._sample(m, 4, 1, 11, 50)
torch.cuda.synchronize(); dt=time.time()-t0
print(prec, "50 steps %.3f s" % dt)
print(" vs f64 :", ["%.2e" % rmse(inter[int(k)]["xt"], gd["xs"][j]) for j, k in enumerate(gd["kept"])])
print(" vs fp32:", ["%.2e" % rmse(inter[int(k)]["xt"], g32["xs"][j]) for j, k in enumerate(gd["kept"])])
