"""CPU-only checks of the C-ABI library and of the host-side mirror (no GPU compute calls)."""
import ctypes as C
import os
import re

import numpy as np
import pytest
import torch

from helpers import ROOT, close

import fab_torch_amd as fa
from fab_torch_amd import _lib
from oracle import ais as oais
from oracle import flow as oflow


def header_symbols():
    txt = open(os.path.join(ROOT, "include", "fabhip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(fabhip_[a-z_0-9]+)\s*\(", txt)))


def test_library_loads_and_exports_every_declared_symbol():
    lib = _lib.load()
    syms = header_symbols()
    assert len(syms) >= 20
    for s in syms:
        assert hasattr(lib, s), f"libfabhip.so does not export {s}"
    assert sorted(_lib.SYMBOLS) == syms, "fab_torch_amd/_lib.py SYMBOLS out of sync with include/fabhip.h"
    assert lib.fabhip_version() >= 100
    assert b"workspace" in lib.fabhip_strerror(-4)


def test_abi_revision_and_developer_switches_without_gpu():
    """The ABI revision of the header, of both bindings and of the library agree; every developer switch of the header exists in
    the library with the default the header documents (the round-4 paths - one-launch phase tails, the in-kernel step-size rule,
    fused spline leapfrogs - are ON by default) and can be set / restored; an unknown key is rejected."""
    from fab_torch_amd import _ops
    lib = _lib.load()
    txt = open(os.path.join(ROOT, "include", "fabhip.h")).read()
    rev = int(re.search(r"#define FABHIP_ABI_VERSION (\d+)", txt).group(1))
    assert lib.fabhip_version() == rev == _lib.ABI_VERSION == _ops.ABI_VERSION
    keys = dict(re.findall(r"#define (FABHIP_OPT_[A-Z0-9_]+) (\d+)", txt))
    count = int(keys.pop("FABHIP_OPT_COUNT"))
    assert sorted(int(v) for v in keys.values()) == list(range(count)) == list(range(12))
    defaults = {"FABHIP_OPT_TILE_SHAPE": 0, "FABHIP_OPT_R4_STREAM": 2, "FABHIP_OPT_SCAN_VARIANT": 3, "FABHIP_OPT_SYSTEMATIC_VARIANT": 1,
                "FABHIP_OPT_SPLINE_STAGED": 0, "FABHIP_OPT_TIMELINE": 0, "FABHIP_OPT_SPLINE_MFMA": 0, "FABHIP_OPT_SPLINE_LEAP": 1,
                "FABHIP_OPT_FUSED_TAIL": 1, "FABHIP_OPT_ADAPT_FOLD": 1, "FABHIP_OPT_PGRAD": 1, "FABHIP_OPT_TAPE_TILES": 0}
    for name, key in keys.items():
        env = name.replace("FABHIP_OPT_", "FABHIP_").replace("TILE_SHAPE", "TILE")
        if env in os.environ:
            continue                                                   # (the library read the developer's override at load time)
        assert lib.fabhip_get_option(int(key)) == defaults[name], name
    assert (_ops.OPT_FUSED_TAIL, _ops.OPT_ADAPT_FOLD) == (int(keys["FABHIP_OPT_FUSED_TAIL"]), int(keys["FABHIP_OPT_ADAPT_FOLD"]))
    prev = lib.fabhip_set_option(int(keys["FABHIP_OPT_ADAPT_FOLD"]), 0)
    assert prev == 1 and lib.fabhip_get_option(int(keys["FABHIP_OPT_ADAPT_FOLD"])) == 0
    lib.fabhip_set_option(int(keys["FABHIP_OPT_ADAPT_FOLD"]), prev)
    assert lib.fabhip_set_option(count, 1) < 0 and lib.fabhip_get_option(-1) < 0
    assert {"FABHIP_AIS_INIT = 1", "FABHIP_AIS_FINISH = 2", "FABHIP_AIS_CONTINUE = 4"} <= set(re.findall(r"FABHIP_AIS_[A-Z]+ = \d", txt))


def test_geometry_queries_and_argument_validation_without_gpu():
    lib = _lib.load()
    n = lib.fabhip_flow_packed_floats(32, 10, 320)
    assert n > 10 * (16 * 320 + 320 * 320 + 320 * 32) * 2            # both orientations, padded
    assert lib.fabhip_flow_packed_floats(65, 2, 32) == -1              # dim beyond compiled limit
    assert lib.fabhip_flow_packed_floats(32, 2, 513) == -1
    assert lib.fabhip_hmc_workspace_bytes(1024, 32, 1) > 0
    assert lib.fabhip_hmc_workspace_bytes(1024, 32, 2) > lib.fabhip_hmc_workspace_bytes(1024, 32, 1)
    assert lib.fabhip_ais_workspace_bytes(1024, 32, 1) > lib.fabhip_hmc_workspace_bytes(1024, 32, 1)
    assert lib.fabhip_resample_workspace_bytes(1 << 20) >= 8 * (1 << 20)
    # null / bad arguments are rejected before any launch
    assert lib.fabhip_flow_log_prob(None, None, None, None, 4, None) == -1
    f = _lib.Flow(32, 2, 64, None)
    assert lib.fabhip_flow_sample(C.byref(f), None, None, None, 4, None) == -1
    assert lib.fabhip_gather_rows(None, None, None, 1, 1, None) == -1
    assert lib.fabhip_ess_logz(None, 4, None, 4.0, None, None, 0, None) == -1


def test_training_path_layout_matches_the_parameter_shapes_without_gpu():
    """fabhip_flow_grad_layout / _grad_views: the flat gradient image has exactly one slot per trainable scalar, in
    the shapes of the normflows-compatible parameters; tape sizes grow with the batch; bad arguments rejected."""
    lib = _lib.load()
    for D, K, nodes in [(32, 10, 10), (6, 8, 40), (5, 3, 4), (2, 4, 40)]:
        flow = fa.RealNVP(D, K, nodes)
        n = lib.fabhip_flow_grad_floats(D, K, D * nodes)
        tensors = flow._grad_tensors()
        assert n == sum(p.numel() for p in tensors) == sum(p.numel() for p in flow.parameters())
        views = flow._grad_views(torch.arange(n, dtype=torch.float32))
        assert [tuple(v.shape) for v in views] == [tuple(p.shape) for p in tensors]
        covered = torch.cat([v.reshape(-1) for v in views])
        assert torch.equal(covered, torch.arange(n, dtype=torch.float32))          # disjoint, complete, ordered
        t1 = lib.fabhip_flow_tape_bytes(D, K, D * nodes, 16)
        t2 = lib.fabhip_flow_tape_bytes(D, K, D * nodes, 17)
        assert 0 < t1 < t2 == lib.fabhip_flow_tape_bytes(D, K, D * nodes, 32)      # rows padded to the 16-chain tile
    assert lib.fabhip_flow_grad_floats(65, 2, 32) == -1
    assert lib.fabhip_flow_tape_bytes(65, 2, 32, 16) == 0
    assert lib.fabhip_flow_grad_layout(32, 10, 320, None) == -1
    f = _lib.Flow(32, 2, 64, None)
    assert lib.fabhip_flow_log_prob_tape(C.byref(f), None, None, None, 4, None, 0, None) == -1
    assert lib.fabhip_flow_param_grad(None, None, None, 0, None, 4, None, None) == -1
    assert lib.fabhip_adam_clip_step(None, None, None, None, 4, 1e-3, 0.9, 0.999, 1e-8, None, 1.0, None, None, 0,
                                     None) == -1
    assert lib.fabhip_adam_workspace_bytes(1000) > 0


def test_anneal_coefficients_match_the_reference_formulas():
    lib = _lib.load()
    for beta in (0.0, 0.2, 1 / 3, 1.0):
        for alpha in (2.0, 0.5):
            a = _lib.Anneal()
            lib.fabhip_anneal_coefs(beta, alpha, 0, C.byref(a))
            assert a.c_q == np.float32((1 - beta) + beta * (1 - alpha)) and a.c_p == np.float32(beta * alpha)
            assert a.g_q == a.c_q and a.g_p == np.float32(2 * beta)       # base.py:116 quirk
            lib.fabhip_anneal_coefs(beta, alpha, 1, C.byref(a))
            assert a.c_q == np.float32(1 - beta) and a.c_p == np.float32(beta) and a.g_p == a.c_p


def test_beta_schedule_and_constructor_contracts():
    flow = fa.RealNVP(6, 2, 5)
    target = fa.ManyWellEnergy(6, use_gpu=False)
    hmc = fa.HamiltonianMonteCarlo(4, 6, flow.log_prob, target.log_prob, alpha=2.0, p_target=False, epsilon=1.0)
    assert hmc.uses_grad_info and set(hmc.state_dict()) == {"common_epsilon", "epsilons", "mass_vector"}
    np.testing.assert_allclose(hmc.epsilons.numpy(), 0.9) ; np.testing.assert_allclose(hmc.common_epsilon.numpy(), 0.1)
    met = fa.Metropolis(4, 6, flow.log_prob, target.log_prob, n_updates=3, alpha=2.0, max_step_size=5.0, min_step_size=1.0)
    assert not met.uses_grad_info and set(met.state_dict()) == {"noise_scalings"}
    np.testing.assert_allclose(met.noise_scalings[0].numpy(), [5.0, 3.0, 1.0])
    met.set_eval_mode(True)
    assert met.eval_mode is False           # the reference inverts the flag (metropolis.py:39-41)
    for M in (1, 4, 8):
        for sp in ("linear", "geometric"):
            ais = fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, False, 2.0, M, sp)
            np.testing.assert_array_equal(ais.B_space.numpy(), oais.beta_schedule(M, sp).numpy())
    with pytest.raises(AssertionError):
        fa.AnnealedImportanceSampler(flow, target.log_prob, hmc, p_target=False, alpha=None)
    assert flow.event_shape == (6,)


def test_hot_path_fails_loudly_without_a_gpu():
    flow = fa.RealNVP(6, 2, 5).requires_grad_(False)
    x = torch.randn(4, 6)
    if not torch.cuda.is_available():
        with pytest.raises(_lib.FabhipError):
            flow.log_prob(x)
        with pytest.raises(_lib.FabhipError):
            fa.ManyWellEnergy(6, use_gpu=False).log_prob(x)
        with pytest.raises(_lib.FabhipError):
            fa.effective_sample_size(torch.randn(8))


def test_state_dict_is_normflows_compatible_and_cpu_calls_fail_loudly():
    torch.manual_seed(0)
    nf = oflow.make_realnvp(6, 3, 5)
    oflow.randomize_last_layers(nf, 0.1, 1)
    flow = fa.RealNVP(6, 3, 5)
    assert set(flow._nf_model.state_dict()) == set(nf.state_dict())
    assert all(k.startswith("_nf_model.") for k in flow.state_dict())
    flow._nf_model.load_state_dict(nf.state_dict())
    from fab_torch_amd._lib import FabhipError
    x = torch.randn(16, 6)
    for call in (lambda: flow.log_prob(x), lambda: flow.sample_and_log_prob((16,)),
                 lambda: flow.sample_and_log_prob((16,), eps=torch.randn(16, 6))):
        with pytest.raises(FabhipError):              # parameters on the CPU / x on the CPU: no ATen fallback
            call()


def test_spline_flow_module_has_normflows_keys_and_fails_loudly_on_the_cpu():
    import math
    from oracle import spline as osp
    tb = torch.full((10,), 5.0); tb[[2, 7]] = math.pi
    of = osp.make_circular_coupled_flow(10, 4, 32, (2, 7), tb, seed=3)
    hf = fa.CircularCoupledRQSFlow(10, 4, 32, (2, 7), tb, seed=3)
    assert set(hf._nf_model.state_dict()) == set(of.state_dict())
    hf._nf_model.load_state_dict(of.state_dict())
    assert hf.event_shape == (10,)
    lib = _lib.load()
    assert lib.fabhip_spline_packed_floats(60, 12, 256) > 12 * 2 * (30 * 256 + 2 * 256 * 256 + 256 * 750)
    assert lib.fabhip_spline_packed_floats(65, 2, 64) == -1 and lib.fabhip_spline_packed_floats(60, 2, 257) == -1
    # training tape layout (host-only arithmetic): 3 [B][64] + 6 [B][Wp] + [B][NFP] + [B][64 * 25] per layer
    import ctypes
    lay = (ctypes.c_int64 * 16)()
    assert lib.fabhip_spline_tape_layout(60, 12, 256, 1024, lay) == 0
    Wp, NFP = lay[13], lay[14]
    assert (Wp, NFP, lay[15]) == (256, 768, 1600) and lay[1] == 1024 * (3 * 64 + 6 * Wp + NFP + 1600)
    assert lay[0] == 12 * lay[1] and list(lay[2:13]) == sorted(lay[2:13])
    assert lib.fabhip_spline_tape_layout(65, 2, 64, 8, lay) != 0
    with pytest.raises(_lib.FabhipError):
        hf.log_prob(torch.randn(4, 10))
    with pytest.raises(NotImplementedError):
        fa.CircularCoupledRQSFlow(10, 4, 32, (2, 7), tb, num_bins=4)
