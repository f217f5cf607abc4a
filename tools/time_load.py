"""Where the model-ready time goes: synthetic ckpt write, FP UNet + weights, get_qmodel (wrap +
load_cali_model), prepare_slots (plan + pack every slot), graph capture — wall clock per phase and a cProfile of the
host side.  usage: python tools/time_load.py [arch=sd] [res=64] [slots=50] [profile=0]"""
import cProfile, os, pstats, sys, tempfile, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from dgq_amd import synth
from dgq_amd.runtime import build_synthetic_qnn

arch = sys.argv[1] if len(sys.argv) > 1 else "sd"
res = int(sys.argv[2]) if len(sys.argv) > 2 else 64
slots = int(sys.argv[3]) if len(sys.argv) > 3 else 50
prof = len(sys.argv) > 4 and sys.argv[4] == "1"
cfg = dict(wbits=4, abits=8, use_aq=True, G=16, log=True, rt=True, sp=True, time_aware=True, steps=slots if arch == "sd" else 4)
os.environ["DGQ_BUILD_TIMING"] = "1"
d = tempfile.mkdtemp()
pr = cProfile.Profile() if prof else None
t0 = time.time()
if pr:
    pr.enable()
qnn, path = build_synthetic_qnn(arch, cfg, res, 2 if arch == "sd" else 1, slots if arch == "sd" else [0, 3], ckpt_dir=d)
torch.cuda.synchronize()
t1 = time.time()
qnn.prepare_slots()
torch.cuda.synchronize()
t2 = time.time()
if pr:
    pr.disable()
print("build_synthetic_qnn %.1f s (ckpt %.1f GB), prepare_slots %.1f s, total %.1f s" % (t1 - t0, os.path.getsize(path) / 1e9, t2 - t1, t2 - t0))
import resource
print("host max RSS %.1f GB, device memory %.1f GB" % (resource.getrusage(resource.RUSAGE_SELF).ru_maxrss / 1e6, torch.cuda.memory_allocated() / 1e9))
if pr:
    pstats.Stats(pr).sort_stats("cumulative").print_stats(45)
