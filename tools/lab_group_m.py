import os, sys, json, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from raft_amd import CorrBlock
dev = torch.device("cuda:0")
def ev_time(fn, reps=10, warm=3):
    for _ in range(warm): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / reps
res = {}
for tag, (H, W) in {"sintel": (55, 128), "1080p": (136, 240)}.items():
    a = torch.randn(1, 256, H, W, device=dev); b = torch.randn(1, 256, H, W, device=dev)
    for gm in (8, 4, 2, 1, 16, 32):
        os.environ["RC_GROUP_M"] = str(gm)
        with torch.no_grad():
            res[f"{tag}_bf16_gm{gm}"] = round(ev_time(lambda: CorrBlock(a, b, 4, 4, precision="bf16")))
            if gm in (8, 4, 16): res[f"{tag}_fp32_gm{gm}"] = round(ev_time(lambda: CorrBlock(a, b, 4, 4)))
print(json.dumps(res))
