"""BUILD-CONTAINER ONLY (imports /root/reference): the imported reference AIS (fab/sampling_methods/ais.py + hmc.py,
with the oracle RealNVP as its base_distribution plug-in) timed beside the oracle restatement (oracle/ais.py) on the
SAME inputs / thread count, to show that the `cpu_baseline` the bench reports (kind "port": the oracle on the GPU box's
host cores) is a fair stand-in for the reference's CPU path.  Writes profiles/r2/cpu_reference_vs_oracle.json."""
import json, os, sys, time, types
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
for name in ["wandb", "normflows", "nflows", "nflows.flows"]:
    sys.modules[name] = types.ModuleType(name)
sys.modules["normflows"].NormalizingFlow = object
sys.modules["nflows"].flows = sys.modules["nflows.flows"]
sys.modules["nflows.flows"].Flow = object
sys.path.insert(0, "/root/reference")
from fab import AnnealedImportanceSampler, HamiltonianMonteCarlo          # noqa: E402
from fab.target_distributions.many_well import ManyWellEnergy            # noqa: E402
from oracle import ais as oais, flow as oflow, targets as otgt            # noqa: E402

D, K, NODES, B, M, L = 32, 10, 10, 1024, 8, 5
torch.set_num_threads(8)
torch.manual_seed(0)
nf = oflow.make_realnvp(D, K, NODES)
oflow.randomize_last_layers(nf, 0.01, 1)


class Base:
    def sample_and_log_prob(self, shape):
        with torch.no_grad():
            return nf.sample_eps(torch.randn(shape[0], D))
    def log_prob(self, x):
        return nf.log_prob(x)
    def sample(self, shape):
        return self.sample_and_log_prob(shape)[0]
    @property
    def event_shape(self):
        return (D,)


def timeit(fn, n=3):
    fn()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    return (time.perf_counter() - t0) / n


target = ManyWellEnergy(dim=D, use_gpu=False)
hmc = HamiltonianMonteCarlo(n_ais_intermediate_distributions=M, dim=D, base_log_prob=nf.log_prob,
                            target_log_prob=target.log_prob, alpha=2.0, p_target=False, epsilon=0.2, n_outer=1, L=L)
ais = AnnealedImportanceSampler(Base(), target.log_prob, hmc, p_target=False, alpha=2.0, n_intermediate_distributions=M)
t_ref = timeit(lambda: ais.sample_and_log_weights(B))

ot = otgt.ManyWell(D)
ohmc = oais.HMC(M, D, nf.log_prob, ot.log_prob, alpha=2.0, p_target=False, epsilon=0.2, L=L)
oa = oais.AIS(lambda e: tuple(t.detach() for t in nf.sample_eps(e)), nf.log_prob, ot.log_prob, ohmc, False, 2.0, M)
t_or = timeit(lambda: oa.sample_and_log_weights(torch.randn(B, D), torch.randn(M, 1, B, D), torch.empty(M, 1, B).exponential_()))
out = {"workload": "ManyWell-32, RealNVP 10x(16-320-320-32)+InvAffine, 1024 chains, M=8, HMC L=5, fp32", "threads": 8,
       "host": "build container (no GPU)", "reference_sec_per_call": t_ref, "oracle_sec_per_call": t_or,
       "reference_samples_per_s": B / t_ref, "oracle_samples_per_s": B / t_or, "oracle_over_reference": t_or / t_ref}
os.makedirs(os.path.join(ROOT, "profiles", "r2"), exist_ok=True)
json.dump(out, open(os.path.join(ROOT, "profiles", "r2", "cpu_reference_vs_oracle.json"), "w"), indent=1)
print(json.dumps(out))
