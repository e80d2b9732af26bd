"""Which HIP runtimes end up in the process when libgaml_hip.so and torch are loaded in either order."""
import sys
sys.path.insert(0, ".")
order = sys.argv[1]


def maps():
    seen = set()
    for l in open("/proc/self/maps"):
        p = l.split()[-1]
        if ("amdhip" in p or "hsa-runtime" in p) and p not in seen:
            seen.add(p)
    return sorted(seen)


if order == "ours_first":
    from gaml_amd import api
    print("after ours:", maps(), flush=True)
    import torch
    print("after torch:", maps(), flush=True)
    print("torch sees gpu:", torch.cuda.is_available(), flush=True)
else:
    import torch
    print("after torch:", maps(), flush=True)
    print("torch sees gpu:", torch.cuda.is_available(), flush=True)
    from gaml_amd import api
    print("after ours:", maps(), flush=True)
try:
    c = api.Context()
    print("context ok", flush=True)
except Exception as e:
    print("context FAILED:", e, flush=True)
