"""CPU oracle for the fab-torch AIS / flow-density hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``oracle/`` is imported by the
product package ``fab_torch_amd``; only ``tests/``, ``__graft_entry__.smoke()``
and the ``cpu_baseline`` leg of ``bench.py`` may use it, and only as the
checker / reported CPU baseline.

It is a plain PyTorch-CPU restatement (eager ops, one ``autograd.grad`` per
leapfrog, exactly the reference's structure) of the reference algorithm:

* AIS driver ............ fab/sampling_methods/ais.py:53-105
* Point / annealed density fab/sampling_methods/base.py:7-124
* HMC .................... fab/sampling_methods/transition_operators/hmc.py:105-202
* Metropolis ............. fab/sampling_methods/transition_operators/metropolis.py:51-74
* ESS .................... fab/utils/numerical.py:18-23
* ManyWell / GMM ......... fab/target_distributions/{double_well,many_well,gmm}.py
* RealNVP flow ........... third-party ``normflows`` (unpinned in the reference's
  requirements.txt:3, absent from /root/reference and from this image);
  restated from its published architecture as used by
  experiments/make_flow/make_normflow_model.py:11-30,82-96.

Parity pinning:
* everything except the flow arithmetic is pinned against the *imported*
  reference (tests/golden/make_golden.py runs the reference's own
  AnnealedImportanceSampler / HamiltonianMonteCarlo / Metropolis / targets /
  ESS / resample in this container and stores inputs+outputs as fixtures);
* the RealNVP arithmetic is "parity unpinned" by the reference (normflows is
  not available offline and the reference's only flow test asserts shapes);
  it is pinned by self-consistency (inverse∘forward, autograd log-det vs
  Jacobian slogdet, finite differences).

All randomness is an explicit *input* (noise tensors), never drawn inside.
"""
