"""ATen (stock PyTorch-ROCm) expression of the RealNVP density on the GPU — TEST INFRASTRUCTURE: the "plain PyTorch
fp32 reference of the same op" next to the CPU oracle, used where a GPU-resident differentiable reference is handy
(parameter-gradient comparisons at sizes the CPU oracle would take minutes for).  Never imported by the product."""
import math

import torch


def log_prob(flow, x):
    """flow: fab_torch_amd.RealNVP (its nn.Parameters are used directly, so autograd reaches them)."""
    q0 = flow._nf_model.q0
    relu = torch.nn.functional.relu
    log_q = torch.zeros(len(x), dtype=x.dtype, device=x.device)
    z = x
    for l1, l2, l3, aff in reversed(list(flow._layers())):
        z = z @ aff.assemble()
        log_q = log_q + torch.sum(aff.log_S)
        z1, z2 = z[:, :flow.d], z[:, flow.d:]
        prm = l3(relu(l2(relu(l1(z1)))))
        shift, scale = prm[:, 0::2], prm[:, 1::2]
        z2 = (z2 - shift) * torch.exp(-scale)
        log_q = log_q - torch.sum(scale, dim=1)
        z = torch.cat([z1, z2], 1)
    base = -0.5 * flow.dim * math.log(2 * math.pi) - torch.sum(
        q0.log_scale + 0.5 * torch.pow((z - q0.loc) / torch.exp(q0.log_scale), 2), 1)
    return log_q + base
