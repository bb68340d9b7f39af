import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["NCCL_DEBUG"] = "INFO"
if len(sys.argv) > 1 and sys.argv[1] == "torch":
    import torch
import kubernetes_autoscaler_amd as kaa
print("HSA_ENABLE_IPC_MODE_LEGACY", os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY"))
try:
    m = kaa.MultiContext([0], use_rccl=True)
    print("ok", m.info())
    m.close()
except Exception as e:
    print("FAILED", e)
print([l.split()[-1] for l in open("/proc/self/maps") if "rccl" in l or "amdhip" in l][::4])
