"""Shared by oracle/make_trajectory_golden.py and tests/test_gpu_training.py: the training-trajectory cases (VERDICT r04
row n2) — which model (MODEL_CASES entry: class, kwargs, weight seed) is trained on which deterministic batches."""
from dig_amd.synthetic import make_batch

STEPS, LR, NB, P_FORCE = 30, 5e-4, 6, 100.0         # run.py:47 defaults (lr 5e-4, p 100); 30 Adam steps over 6 rotating batches

TRAJ = {
    # case: batch generator of the NB training batches (seeds 500 + k) and of the held-out batch (seed 599)
    'spherenet_tiny': dict(n_min=5, n_max=9, rho=0.08, cutoff=5.0, batch=4),
    'schnet_cfg1_b32': dict(n_min=9, n_max=29, rho=0.08, cutoff=10.0, batch=32),
    # the models the metric is quoted on (BASELINE configs 2, 3, and ComENet at its QM9 defaults)
    'spherenet_default_b32': dict(n_min=9, n_max=29, rho=0.08, cutoff=5.0, batch=32),
    'dimenetpp_force_md17_b8': dict(n_min=21, n_max=21, rho=0.09, cutoff=5.0, batch=8, with_force=True),
    'comenet_default_b8': dict(n_min=9, n_max=29, rho=0.08, cutoff=8.0, batch=8),
}


def traj_batches(case):
    t = TRAJ[case]
    mk = lambda seed: make_batch(t['batch'], t['n_min'], t['n_max'], t['rho'], t['cutoff'], seed=seed,
                                 with_force=t.get('with_force', False))
    return [mk(500 + k) for k in range(NB)], mk(599)
