"""Dev tool: fabhip_resample_multinomial (scan + per-draw CDF search) at large N for three weight distributions."""
import json, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import fab_torch_amd as fa
from tools.bench_extra import ev_time
DEV = torch.device("cuda", 0)
out = {}
for N in (1 << 20, 1 << 24, 1 << 26):
    g = torch.Generator(device=DEV).manual_seed(0)
    u = torch.rand(N, dtype=torch.float64, device=DEV, generator=g)
    for name, lw in (("normal_sigma3", torch.randn(N, device=DEV, generator=g) * 3),
                     ("normal_sigma0.5", torch.randn(N, device=DEV, generator=g) * 0.5),
                     ("uniform_weights", torch.zeros(N, device=DEV))):
        t = ev_time(lambda: fa.multinomial_indices(lw, u=u), n=5, warm=2)
        idx = fa.multinomial_indices(lw, u=u)
        out[f"N={N} {name}"] = {"ms": t * 1e3, "checksum": int(idx.sum().item()), "max": int(idx.max().item())}
print(json.dumps(out, indent=1))
