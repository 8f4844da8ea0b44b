"""bench.py -- PPO update() throughput of the MI355X-native learner path (NatureCNN 84x84x4).

    python bench.py [--gpus N] [--steps K] [--warmup W]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

Both forms work for N > 1: without a torch.distributed.run environment (no WORLD_SIZE) `--gpus N` starts its own N
worker processes -- fresh children, started before this process has made any GPU call -- one rank per GPU over RCCL;
rank 0 prints the JSON line, any failing rank makes the whole command exit non-zero.

Workload (BASELINE.json configs[2], the configuration the metric "PPO ... update samples/sec (NatureCNN
84x84x4) at 1/2/4/8 GPU" is quoted on): PPO, 4 epochs x 4 minibatches, T = 128 steps, 32 env processes PER
GPU (256 processes at 8 GPUs = the top of the config's range), Categorical(6), synthetic uint8-valued frames
stored as fp32 (F6), clip .1, lr 2.5e-4, entropy .01, GAE.  Weak scaling: per-GPU work is fixed.
A "step" is one full `PPO.update(rollouts)`; samples per step = T * N_global * ppo_epoch.

One JSON line is printed by rank 0; see DESIGN.md "Measurement" for how each field is derived.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for p in (ROOT, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd')):
    if p not in sys.path:
        sys.path.insert(0, p)

import torch                                    # noqa: E402
import torch.distributed as dist                # noqa: E402

T_STEPS, PROCS_PER_GPU, ACTIONS = 128, 32, 6
PPO_HP = dict(clip_param=0.1, ppo_epoch=4, num_mini_batch=4, value_loss_coef=0.5, entropy_coef=0.01, lr=2.5e-4,
              eps=1e-5, max_grad_norm=0.5)
MFMA_F32_PEAK_TFLOPS = 157.3      # /opt/skills/guides/MI355X_MICROARCH.md, "Peak FP32 (matrix)"
FWD_BWD_FLOP_PER_SAMPLE = 39.29e6  # 15.28 fwd + 24.01 bwd MFLOP per sample per minibatch pass


class Discrete(object):
    def __init__(self, n):
        self.n = n


class Box(object):
    def __init__(self, dim):
        self.shape = (dim,)


def synth_rollouts(T, N, device, seed):
    """SURVEY.md 8(d) synthetic inputs, generated on `device` from a fixed seed (identical on every rank)."""
    g = torch.Generator(device=device).manual_seed(seed)
    ro = {}
    ro['obs'] = torch.randint(0, 256, (T + 1, N, 4, 84, 84), device=device, generator=g, dtype=torch.int32).float()
    ro['rewards'] = torch.randn(T, N, 1, device=device, generator=g)
    ro['value_preds'] = torch.randn(T + 1, N, 1, device=device, generator=g)
    ro['action_log_probs'] = 0.1 * torch.randn(T, N, 1, device=device, generator=g) - 1.8
    ro['actions'] = torch.randint(0, ACTIONS, (T, N, 1), device=device, generator=g)
    ro['masks'] = (torch.rand(T + 1, N, 1, device=device, generator=g) < 0.99).float()
    ro['bad_masks'] = (torch.rand(T + 1, N, 1, device=device, generator=g) < 0.995).float()
    ro['next_value'] = torch.randn(N, 1, device=device, generator=g)
    return ro


def build_agent(device, seed):
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.model import Policy
    torch.manual_seed(seed)                       # reference initialisers (orthogonal_) on the CPU generator
    pol = Policy((4, 84, 84), Discrete(ACTIONS))
    pol.to(device)
    return pol, algo.PPO(pol, **PPO_HP)


def fill_storage(ro, device):
    from a2c_ppo_acktr.storage import RolloutStorage
    T, N = ro['rewards'].shape[:2]
    rs = RolloutStorage(1, 1, (1,), Discrete(ACTIONS), 1)     # tiny CPU allocation, tensors replaced below
    rs.num_steps = T
    for k in ('obs', 'rewards', 'value_preds', 'action_log_probs', 'actions', 'masks', 'bad_masks'):
        setattr(rs, k, ro[k])
    rs.returns = torch.zeros(T + 1, N, 1, device=device)
    rs.recurrent_hidden_states = torch.zeros(T + 1, N, 1, device=device)
    return rs


def cpu_baseline(ro_cpu, seed, max_updates=12, budget_s=12.0):
    """The oracle (CPU restatement of the reference's PPO.update, torch CPU ops) on the per-GPU workload:
    full updates of T=128 x 32 processes, 4 epochs x 4 minibatches (bounded sample, ~10-30 s)."""
    from collections import OrderedDict
    from oracle import algo_ref, returns_ref
    from a2c_ppo_acktr.model import Policy
    torch.manual_seed(seed)
    pol = Policy((4, 84, 84), Discrete(ACTIONS))
    P = OrderedDict((k, v.detach().clone()) for k, v in pol.state_dict().items())
    ro = {k: v.clone() for k, v in ro_cpu.items()}
    ro['returns'] = torch.zeros_like(ro['value_preds'])
    npv = {k: ro[k].numpy() for k in ('rewards', 'value_preds', 'returns', 'masks', 'bad_masks', 'next_value')}
    returns_ref.compute_returns(npv['rewards'], npv['value_preds'], npv['returns'], npv['masks'], npv['bad_masks'],
                                npv['next_value'], True, 0.99, 0.95, True)
    opt = algo_ref.adam_init(P)
    torch.manual_seed(seed + 1)
    T, N = ro['rewards'].shape[:2]
    t0 = time.perf_counter()
    n_upd, losses = 0, None
    while n_upd < max_updates and (n_upd == 0 or time.perf_counter() - t0 < budget_s):   # bounded sample of host work
        res = algo_ref.ppo_update(P, opt, ro, **PPO_HP)
        if losses is None:
            losses = res          # the FIRST update's (value_loss, action_loss, entropy): main() checks the GPU's against it
        n_upd += 1
    dt = time.perf_counter() - t0
    return dict(value=n_upd * T * N * PPO_HP['ppo_epoch'] / dt, unit='samples/s', cores=torch.get_num_threads(),
                kind='port',
                sample='%d PPO.update of T=%d x N=%d (4 epochs x 4 minibatches of %d) on the host, %.1f s'
                       % (n_upd, T, N, T * N // PPO_HP['num_mini_batch'], dt)), losses


def _profile_traffic(op, stats_kind, traffic_kind):
    """HBM-side bytes per launch of operator `op` from the newest committed counter summary
    profiles/rN_<traffic_kind>_<tag>.json (separate rocprofv3 --pmc passes of this command, tools/profile_round.sh +
    tools/pmc_traffic.py) -- refused (None, with the reason) when a NEWER kernel-stats summary rN_<stats_kind>_<tag>.csv
    exists, i.e. when the kernels were profiled again after the counters were taken.  Returns (bytes or None, source)."""
    import glob
    import re

    def newest(pattern, rx):
        best = None
        for f in glob.glob(os.path.join(ROOT, 'profiles', pattern)):
            m = re.match(rx, os.path.basename(f))
            if m and (best is None or (int(m.group(1)), m.group(2)) > best[0]):
                best = ((int(m.group(1)), m.group(2)), f)
        return best
    try:
        tr = newest('r*_%s_*.json' % traffic_kind, r'r(\d+)_%s_(\w+)\.json$' % traffic_kind)
        ks = newest('r*_%s_*.csv' % stats_kind, r'r(\d+)_%s_(\w+)\.csv$' % stats_kind)
        if tr is None:
            return None, 'no profiles/r*_%s_*.json' % traffic_kind
        tags = 'counters r%d tag %s, kernel stats %s' % (tr[0][0], tr[0][1], 'r%d tag %s' % ks[0] if ks else 'none')
        if ks is not None and ks[0] > tr[0]:
            return None, 'refused: %s are newer than the counter passes (%s)' % (os.path.basename(ks[1]), tags)
        val = json.load(open(tr[1])).get(op, {}).get('traffic_bytes')
        return val, 'profiles/%s (2*FETCH_SIZE + WRITE_SIZE per launch, separate --pmc passes; %s)' % (
            os.path.basename(tr[1]), tags)
    except Exception as e:
        return None, 'unreadable: %r' % (e,)


def _timed_updates(agent, rs, n, seed0):
    """ms of each of n consecutive update() calls (device-synchronised around every call)."""
    out = []
    for i in range(n):
        torch.manual_seed(seed0 + i)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        agent.update(rs)
        torch.cuda.synchronize()
        out.append((time.perf_counter() - t0) * 1e3)
    return out


def side_metrics(device):
    """The other figures BASELINE.json's metric names, on one GPU (SURVEY.md 8(d) definitions):
    A2C update (configs[1]: 16 procs x 5 steps), ACKTR update + KFAC step ms (configs[3]: 32 procs x 20 steps,
    Ts=1, Tf=10), and the GAE scan at the large-scan stress size of configs[4] (2048 steps x 1024 procs)."""
    import statistics
    from a2c_ppo_acktr import algo
    from a2c_ppo_acktr.model import Policy
    from a2c_ppo_acktr.storage import RolloutStorage
    from a2c_ppo_acktr import ops
    out = {}
    # ---- A2C, B = 80
    T, N = 5, 16
    ro = synth_rollouts(T, N, device, seed=21)
    torch.manual_seed(8)
    pol = Policy((4, 84, 84), Discrete(ACTIONS)); pol.to(device)
    agent = algo.A2C_ACKTR(pol, 0.5, 0.01, lr=7e-4, eps=1e-5, alpha=0.99, max_grad_norm=0.5)
    rs = fill_storage(ro, device)
    rs.compute_returns(ro['next_value'], False, 0.99, 0.95, False)
    _timed_updates(agent, rs, 3, 400)
    ms = _timed_updates(agent, rs, 20, 410)
    out['a2c_update'] = dict(workload='A2C_ACKTR.update NatureCNN, 16 procs x 5 steps (B=80), RMSprop',
                             ms_median=round(statistics.median(ms), 3),
                             samples_per_s=round(T * N / (statistics.median(ms) * 1e-3), 1))
    # ---- ACKTR, B = 640: 23 updates = eigendecompositions at optimizer steps 0, 10, 20
    T, N = 20, 32
    ro = synth_rollouts(T, N, device, seed=22)
    torch.manual_seed(9)
    pol = Policy((4, 84, 84), Discrete(ACTIONS)); pol.to(device)
    agent = algo.A2C_ACKTR(pol, 0.5, 0.01, acktr=True)
    rs = fill_storage(ro, device)
    rs.compute_returns(ro['next_value'], False, 0.99, 0.95, False)
    opt = agent.optimizer
    step_ms, inner = [], opt.step

    def timed_step(*a, **k):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        r = inner(*a, **k)
        torch.cuda.synchronize()
        step_ms.append(((opt.steps - 1) % opt.Tf == 0, (time.perf_counter() - t0) * 1e3))
        return r
    _timed_updates(agent, rs, 2, 500)                       # warm-up (includes the step-0 eigendecomposition)
    ms = _timed_updates(agent, rs, 21, 510)                 # optimizer steps 2..22: eig at 10 and 20
    # KFACOptimizer.step() on its own: launch by launch (inside the replayed graph it has no host-visible boundary)
    agent.use_graph = False
    opt.step = timed_step
    _timed_updates(agent, rs, 21, 540)                      # optimizer steps 23..43: eig at 30 and 40
    opt.step = inner
    # one more update launch by launch with HIP events around the KFAC GEMM-class operators (factor SYRKs on the
    # symmetric-half flop count R*D*(D+1) of SURVEY 8(d), preconditioner GEMMs 4*OUT*IN*(OUT+IN))
    from a2c_ppo_acktr import ops
    ops.profiler = {}
    _timed_updates(agent, rs, 1, 570)
    kprof = {k: v for k, v in ops.profile_summary().items() if k.startswith('kfac_')}
    ops.profiler = None
    agent.use_graph = True
    eig = [m for is_eig, m in step_ms if is_eig]
    plain = [m for is_eig, m in step_ms if not is_eig]
    out['acktr_update'] = dict(workload='A2C_ACKTR(acktr=True).update NatureCNN, 32 procs x 20 steps (B=640), Ts=1 Tf=10',
                               ms_mean_21_updates=round(sum(ms) / len(ms), 3),
                               ms_median_update_without_eig=round(statistics.median(ms), 3),
                               samples_per_s=round(T * N * len(ms) / (sum(ms) * 1e-3), 1),
                               kfac_step_ms=dict(with_eig=round(statistics.median(eig), 3) if eig else None,
                                                 without_eig=round(statistics.median(plain), 3), n_eig=len(eig)))
    if kprof:
        fac = {k: v for k, v in kprof.items() if k.startswith('kfac_cov_a_conv')}
        dom = max(fac.items(), key=lambda kv: kv[1]['total_ms'])
        ach = dom[1]['flops_per_call'] / (dom[1]['avg_ms'] * 1e-3) / 1e12
        ktraffic, ktraffic_src = _profile_traffic(dom[0], 'acktr_kernel_stats', 'acktr_pmc_traffic')
        out['roofline_kfac'] = dict(bound='mfma', kernel=dom[0], achieved=round(ach, 2), peak=MFMA_F32_PEAK_TFLOPS,
                                    unit='TFLOP/s', frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4), traffic=ktraffic,
                                    traffic_source=ktraffic_src,
                                    avg_launch_ms=round(dom[1]['avg_ms'], 4), flop_per_launch=dom[1]['flops_per_call'],
                                    flop_count='symmetric half R*D*(D+1) (SURVEY 8(d)); the kernels compute the upper '
                                               'triangle of 32x32 tiles (stationary accumulators) and mirror it',
                                    operators={k: dict(calls=v['calls'], avg_ms=round(v['avg_ms'], 4),
                                                       tflops=round(v['flops_per_call'] / (v['avg_ms'] * 1e-3) / 1e12, 2))
                                               for k, v in sorted(kprof.items(), key=lambda kv: -kv[1]['total_ms'])})
    # ---- configs[2] at the top of its range on ONE GPU: 256 processes x 128 steps, 4 epochs x 4 minibatches of 8192
    # (the headline line keeps 32 processes per GPU so that 8 GPUs land on this global size)
    T, N = T_STEPS, 256
    ro = synth_rollouts(T, N, device, seed=27)
    pol, agent = build_agent(device, seed=7)
    rs = fill_storage(ro, device)
    rs.compute_returns(ro['next_value'], True, 0.99, 0.95, True)
    _timed_updates(agent, rs, 2, 800)
    ms = _timed_updates(agent, rs, 5, 810)
    med = statistics.median(ms)
    spu = T * N * PPO_HP['ppo_epoch']
    out['ppo_update_256procs'] = dict(workload='PPO.update NatureCNN, 256 procs x 128 steps on one GPU (minibatch 8192)',
                                      ms_median=round(med, 3), samples_per_s=round(spu / (med * 1e-3), 1),
                                      mfma_tflops=round(spu * FWD_BWD_FLOP_PER_SAMPLE / (med * 1e-3) / 1e12, 2),
                                      frac_of_f32_mfma_peak=round(spu * FWD_BWD_FLOP_PER_SAMPLE / (med * 1e-3) / 1e12
                                                                  / MFMA_F32_PEAK_TFLOPS, 4))
    del rs, agent, pol, ro
    torch.cuda.empty_cache()
    # ---- the headline workload with the frames kept as bytes on the device (RolloutStorage(obs_dtype=torch.uint8), not
    # the reference's float32 storage -- which is why it is a side metric): same kernels apart from conv1's loaders,
    # bit-identical results (tests/test_u8_gpu.py), a quarter of the observation traffic
    T, N = T_STEPS, PROCS_PER_GPU
    ro = synth_rollouts(T, N, device, seed=1234)
    ro['obs'] = ro['obs'].to(torch.uint8)
    pol, agent = build_agent(device, seed=7)
    rs = fill_storage(ro, device)
    rs.compute_returns(ro['next_value'], True, 0.99, 0.95, True)
    _timed_updates(agent, rs, 3, 900)
    ms = _timed_updates(agent, rs, 10, 910)
    med = statistics.median(ms)
    spu = T * N * PPO_HP['ppo_epoch']
    ops.profiler = {}
    agent.update(rs)
    k8 = {k: round(v['avg_ms'] * 1e3, 1) for k, v in ops.profile_summary().items() if k.startswith('conv8s4')}
    ops.profiler = None
    out['ppo_update_uint8_obs'] = dict(workload='PPO.update NatureCNN, 32 procs x 128 steps, frames stored as uint8 on the device '
                                                '(minibatch 1024)',
                                       ms_median=round(med, 3), samples_per_s=round(spu / (med * 1e-3), 1),
                                       frac_of_f32_mfma_peak=round(spu * FWD_BWD_FLOP_PER_SAMPLE / (med * 1e-3) / 1e12
                                                                   / MFMA_F32_PEAK_TFLOPS, 4),
                                       conv1_us_by_hip_events=k8)
    del rs, agent, pol, ro
    torch.cuda.empty_cache()
    # ---- GAE scan (20 B per element: GAE + proper time limits): the stress size of configs[4], where the serial
    # chain of 2048 steps on only 1024 lanes is the floor, and a wide rollout where HBM is
    for T, N in ((2048, 1024), (512, 65536)):
        g = torch.Generator(device=device).manual_seed(23)
        rs = RolloutStorage(1, 1, (1,), Discrete(ACTIONS), 1)
        rs.num_steps = T
        rs.rewards = torch.randn(T, N, 1, device=device, generator=g)
        rs.value_preds = torch.randn(T + 1, N, 1, device=device, generator=g)
        rs.returns = torch.zeros(T + 1, N, 1, device=device)
        rs.masks = (torch.rand(T + 1, N, 1, device=device, generator=g) < 0.99).float()
        rs.bad_masks = (torch.rand(T + 1, N, 1, device=device, generator=g) < 0.995).float()
        nv = torch.randn(N, 1, device=device, generator=g)
        for _ in range(3):
            rs.compute_returns(nv, True, 0.99, 0.95, True)
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        ev0.record()
        for _ in range(20):
            rs.compute_returns(nv, True, 0.99, 0.95, True)
        ev1.record()
        torch.cuda.synchronize()
        ms1 = ev0.elapsed_time(ev1) / 20
        out['gae_scan_%dx%d' % (T, N)] = dict(ms=round(ms1, 4), algorithmic_bytes=T * N * 20,
                                              gbps=round(T * N * 20 / (ms1 * 1e-3) / 1e9, 1), hbm_peak_gbps=8000)
        del rs
    # ---- configs[0]: PPO on the Hopper-shaped MLP (obs 11, act 3, T=2048, N=1, 10 epochs x 32 minibatches of 64):
    # launch/latency bound -> microseconds per minibatch step
    T, N = 2048, 1
    g = torch.Generator(device=device).manual_seed(24)
    torch.manual_seed(10)
    pol = Policy((11,), Box(3)); pol.to(device)
    agent = algo.PPO(pol, 0.2, 10, 32, 0.5, 0.0, lr=3e-4, eps=1e-5, max_grad_norm=0.5)
    rs = RolloutStorage(1, 1, (1,), Box(3), 1)
    rs.num_steps = T
    rs.obs = torch.randn(T + 1, N, 11, device=device, generator=g)
    rs.actions = torch.randn(T, N, 3, device=device, generator=g)
    rs.rewards = torch.randn(T, N, 1, device=device, generator=g)
    rs.value_preds = torch.randn(T + 1, N, 1, device=device, generator=g)
    rs.action_log_probs = 0.1 * torch.randn(T, N, 1, device=device, generator=g) - 2
    rs.masks = (torch.rand(T + 1, N, 1, device=device, generator=g) < 0.99).float()
    rs.bad_masks = (torch.rand(T + 1, N, 1, device=device, generator=g) < 0.995).float()
    rs.returns = torch.zeros(T + 1, N, 1, device=device)
    rs.recurrent_hidden_states = torch.zeros(T + 1, N, 1, device=device)
    rs.compute_returns(torch.randn(N, 1, device=device, generator=g), True, 0.99, 0.95, True)
    _timed_updates(agent, rs, 2, 600)
    ms = _timed_updates(agent, rs, 10, 610)
    med = statistics.median(ms)
    out['ppo_mlp_update'] = dict(workload='PPO.update MLPBase obs 11 / DiagGaussian(3), T=2048 x 1 proc, 10 epochs x 32 '
                                          'minibatches of 64 (configs[0])',
                                 ms_median=round(med, 3), us_per_minibatch_step=round(med * 1e3 / 320, 2),
                                 samples_per_s=round(T * N * 10 / (med * 1e-3), 1))
    del rs, agent, pol
    # ---- configs[4]: GAIL on HalfCheetah-shaped data (obs 17, act 6, T=2048 x 1024 procs): discriminator rewards for the
    # whole rollout in one batched call, and the discriminator update against a synthetic gail_experts-format file
    import tempfile
    from a2c_ppo_acktr.algo import gail
    T, N, Ds, Da = 2048, 1024, 17, 6
    g = torch.Generator(device=device).manual_seed(25)
    torch.manual_seed(11)
    disc = gail.Discriminator(Ds + Da, 100, device)
    rs = RolloutStorage(1, 1, (1,), Box(Da), 1)
    rs.num_steps = T
    rs.obs = torch.randn(T + 1, N, Ds, device=device, generator=g)
    rs.actions = torch.randn(T, N, Da, device=device, generator=g)
    rs.rewards = torch.zeros(T, N, 1, device=device)
    rs.value_preds = torch.zeros(T + 1, N, 1, device=device)
    rs.action_log_probs = torch.zeros(T, N, 1, device=device)
    rs.masks = (torch.rand(T + 1, N, 1, device=device, generator=g) < 0.99).float()
    rs.bad_masks = torch.ones(T + 1, N, 1, device=device)
    rs.returns = torch.zeros(T + 1, N, 1, device=device)
    rs.recurrent_hidden_states = torch.zeros(T + 1, N, 1, device=device)
    pr = []
    for i in range(4):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rs.rewards.copy_(disc.predict_reward_batched(rs.obs[:-1], rs.actions, 0.99, rs.masks[:-1]))
        torch.cuda.synchronize()
        pr.append((time.perf_counter() - t0) * 1e3)
    with tempfile.TemporaryDirectory() as td:
        gcpu = torch.Generator().manual_seed(26)
        fn = os.path.join(td, 'trajs_synthetic.pt')
        torch.save({'states': torch.randn(64, 1000, Ds, generator=gcpu), 'actions': torch.randn(64, 1000, Da, generator=gcpu),
                    'rewards': torch.randn(64, 1000, generator=gcpu), 'lengths': torch.full((64,), 1000, dtype=torch.int64)}, fn)
        res = {}
        for ntraj in (4, 64):                       # 4 = main.py's default (one batch of 128 per update); 64 -> 25 batches
            torch.manual_seed(12)
            ds = gail.ExpertDataset(fn, num_trajectories=ntraj, subsample_frequency=20)
            loader = torch.utils.data.DataLoader(ds, batch_size=128, shuffle=True, drop_last=True)
            obsfilt = lambda x, update=False: x
            ms, step_us = [], []
            inner_step = disc._step

            def timed_step(*a, **k):              # HIP events on the launch stream behind every minibatch step
                r = inner_step(*a, **k)
                e = torch.cuda.Event(enable_timing=True)
                e.record()
                evs.append(e)
                return r
            for i in range(6):
                torch.manual_seed(700 + i)
                evs = []
                disc._step = timed_step
                torch.cuda.synchronize()
                t0 = time.perf_counter()
                disc.update(loader, rs, obsfilt)
                torch.cuda.synchronize()
                ms.append((time.perf_counter() - t0) * 1e3)
                disc._step = inner_step
                if len(evs) > 1:                  # steps 2..n on the device time line: the CPU randperm lies before step 1
                    step_us.append(evs[0].elapsed_time(evs[-1]) * 1e3 / (len(evs) - 1))
            res[ntraj] = (statistics.median(ms[2:]), len(loader), statistics.median(step_us[2:]) if step_us else None)
    t0 = time.perf_counter()
    for _ in range(3):
        torch.randperm(T * N)
    rp_ms = (time.perf_counter() - t0) / 3 * 1e3
    out['gail'] = dict(workload='GAIL Discriminator 23->100->100->1 (configs[4]: obs 17, act 6, T=2048 x 1024 procs, batch 128)',
                       predict_reward_rollout_ms=round(min(pr[1:]), 3),
                       predict_reward_samples_per_s=round(T * N / (min(pr[1:]) * 1e-3), 1),
                       update_ms_4_trajectories=round(res[4][0], 3), update_batches_4_trajectories=res[4][1],
                       update_ms_64_trajectories=round(res[64][0], 3), update_batches_64_trajectories=res[64][1],
                       us_per_discriminator_step_host_wall_incl_randperm=round(res[64][0] * 1e3 / res[64][1], 1),
                       # every update draws torch.randperm over all T*N rollout rows on the CPU first (reference
                       # storage.py:119-123 through gail.py:60, same RNG stream): a fixed host cost per update
                       cpu_randperm_ms_per_update=round(rp_ms, 3),
                       # a minibatch step on the device time line: HIP events behind every step of the 25-step update,
                       # (last - first) / 24 -- the permutation is drawn before the first step, so it is not in here
                       us_per_discriminator_step_hip_events=(round(res[64][2], 1) if res[64][2] is not None else None))
    return out


def _free_port():
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(('127.0.0.1', 0))
        return sk.getsockname()[1]


def self_launch(args, argv):
    """`--gpus N` (N > 1) outside torch.distributed.run: start N worker processes of this script, one rank per GPU.
    This process has made NO GPU call (torch.cuda.device_count() does not initialise the runtime on this image) and
    never makes one: it only starts the children, forwards rank 0's stdout (the JSON line) and reaps them.  New in
    this build -- the reference is single device (main.py:39).  Returns the exit code."""
    import subprocess
    n = args.gpus
    if not args.launch_check:
        have = torch.cuda.device_count()
        if have < n:
            sys.stderr.write('bench.py: --gpus %d needs %d GPUs, this machine exposes %d; nothing was started\n'
                             % (n, n, have))
            return 2
    env = dict(os.environ)
    env.update(WORLD_SIZE=str(n), LOCAL_WORLD_SIZE=str(n), MASTER_ADDR='127.0.0.1', MASTER_PORT=str(_free_port()),
               HSA_ENABLE_IPC_MODE_LEGACY='0', ACKTR_BENCH_WORKER='1')
    procs = []
    for r in range(n):
        e = dict(env, RANK=str(r), LOCAL_RANK=str(r))
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + argv, env=e,
                                      stdout=None if r == 0 else sys.stderr))
    deadline = time.monotonic() + float(os.environ.get('ACKTR_BENCH_LAUNCH_TIMEOUT', '1500'))
    rc = 0
    live = list(procs)
    while live:
        for p in list(live):
            c = p.poll()
            if c is None:
                continue
            live.remove(p)
            if c != 0 and rc == 0:
                rc = c if c > 0 else 128 - c
                sys.stderr.write('bench.py: rank %d exited with %d; stopping the other ranks\n' % (procs.index(p), c))
        if rc != 0 or time.monotonic() > deadline:
            if rc == 0:
                rc = 124
                sys.stderr.write('bench.py: ranks still running at the launcher deadline; stopping them\n')
            for p in live:                      # exactly the children started above
                p.terminate()
            for p in live:
                try:
                    p.wait(timeout=20)
                except subprocess.TimeoutExpired:
                    p.kill()
                    p.wait()
            break
        time.sleep(0.05)
    return rc


def dp_mode_string(world, backend_world):
    """What the data-parallel PPO step runs as (a2c_ppo_acktr/algo/ppo.py): part of config.parallelism."""
    g = os.environ.get('ACKTR_DP_GRAPH', '0')
    ov = os.environ.get('ACKTR_DP_OVERLAP', '1') != '0'
    if world == 1 and backend_world is None:
        return 'dp1 (single process, one hipGraph per epoch, no collectives)'
    step = {'0': 'eager launches', '1': 'hipGraph pieces around the collectives',
            'full': 'one hipGraph per epoch with the RCCL all-reduces captured'}.get(g, g)
    return 'dp%d (RCCL communicator of %s ranks; ACKTR_DP_GRAPH=%s: %s; late-bucket all-reduce %s)' % (
        world, backend_world, g, step, 'overlapped with the backward' if ov else 'not overlapped')


def launch_check(rank, world, json_fd):
    """`--launch-check`: the launcher path up to the first device call, on gloo (CPU test of self_launch / the
    torch.distributed.run environment): rendezvous, one SUM all-reduce, barrier, rank 0 prints one JSON line."""
    import datetime
    if os.environ.get('ACKTR_BENCH_TEST_FAIL_RANK') == str(rank):       # tests/test_parallel_cpu.py: a rank that dies
        sys.exit(7)
    dist.init_process_group('gloo', rank=rank, world_size=world, timeout=datetime.timedelta(seconds=60))
    x = torch.tensor([float(rank + 1)])
    dist.all_reduce(x)
    ok = float(x.item()) == world * (world + 1) / 2
    dist.barrier()
    if rank == 0:
        os.write(json_fd, (json.dumps(dict(launch_check=ok, world=dist.get_world_size(), backend='gloo',
                                           sum=float(x.item()))) + '\n').encode())
    dist.destroy_process_group()
    return 0 if ok else 4


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    # defaults: 20 timed updates (~0.2 s) behind 5 warm-up updates -- with 5 / 2 the 50 ms timed window showed +-4 %
    # run-to-run swings on a cold box (clock ramp, one slow replay), far more than any kernel change this round
    ap.add_argument('--steps', type=int, default=20)
    ap.add_argument('--warmup', type=int, default=5)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--no-side-metrics', action='store_true')
    ap.add_argument('--force-dp', action='store_true',
                    help='one rank, but through the data-parallel code path on RCCL (rehearsal of the N>1 path)')
    ap.add_argument('--launch-check', action='store_true',
                    help='rendezvous + one all-reduce on gloo, no device call: tests the N-rank launcher on a CPU box')
    args = ap.parse_args()

    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        sys.exit(self_launch(args, sys.argv[1:]))

    # stdout carries exactly ONE line, the JSON record: everything else that lands on file descriptor 1 -- RCCL and
    # gloo print banners there from C when a communicator is created -- is sent to stderr for the rest of the run
    sys.stdout.flush()
    json_fd = os.dup(1)
    os.dup2(2, 1)
    if args.launch_check:
        sys.exit(launch_check(int(os.environ.get('RANK', '0')), int(os.environ.get('WORLD_SIZE', '1')), json_fd))

    world = int(os.environ.get('WORLD_SIZE', '1'))
    rank = int(os.environ.get('RANK', '0'))
    local_rank = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        sys.stderr.write('bench.py: --gpus %d but the launcher environment says WORLD_SIZE=%d\n' % (args.gpus, world))
        sys.exit(2)
    have = torch.cuda.device_count()
    if local_rank >= have:
        sys.stderr.write('bench.py: rank %d wants cuda:%d, this machine exposes %d GPU(s)\n' % (rank, local_rank, have))
        sys.exit(2)
    rccl_world = None
    if world > 1 or args.force_dp:
        import datetime
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        os.environ.setdefault('MASTER_PORT', '29533')
        os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')
        torch.cuda.set_device(local_rank)
        # bounded rendezvous: a rank that never shows up is a failure record, not a hang
        dist.init_process_group('nccl', rank=rank, world_size=world, device_id=torch.device('cuda', local_rank),
                                timeout=datetime.timedelta(seconds=int(os.environ.get('ACKTR_BENCH_INIT_TIMEOUT', '300'))))
        # the rank count the RCCL communicator itself reports, checked with one SUM all-reduce over it
        probe = torch.ones(1, device='cuda:%d' % local_rank)
        dist.all_reduce(probe)
        rccl_world = int(probe.item())
        if rccl_world != world or dist.get_world_size() != world:
            sys.stderr.write('bench.py: RCCL communicator spans %d ranks, expected %d\n' % (rccl_world, world))
            sys.exit(2)
    device = torch.device('cuda', local_rank)
    torch.cuda.set_device(device)

    from a2c_ppo_acktr import ops
    N_global = PROCS_PER_GPU * world
    ro = synth_rollouts(T_STEPS, N_global, device, seed=1234)
    pol, agent = build_agent(device, seed=7)
    if args.force_dp:
        from a2c_ppo_acktr.parallel import DataParallel
        agent.dp = DataParallel(force_active=True)
    rs = fill_storage(ro, device)
    rs.compute_returns(ro['next_value'], True, 0.99, 0.95, True)

    def barrier():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    losses = None
    for i in range(args.warmup):
        torch.manual_seed(100 + i)
        losses = agent.update(rs)
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        torch.manual_seed(200 + i)
        losses = agent.update(rs)
    barrier()
    dt = time.perf_counter() - t0
    # Per-kernel durations: the timed region above replays a captured hipGraph (no host launches to bracket), so
    # the same update is run once more launch-by-launch with HIP events around every GEMM-class operator on the
    # launch stream (same kernels, shapes, buffers).  rocprofv3 of this command (profiles/) sees both.
    # (run twice, the first pass un-recorded: the eager launches allocate their outputs outside the graph's private
    # pool, and a first-touch hipMalloc inside an event bracket showed up as a 1 ms conv2 forward)
    for _ in range(2):
        ops.profiler = {}
        torch.manual_seed(300)
        agent.update(rs)
    prof = ops.profile_summary()
    ops.profiler = None
    if world > 1:
        tmax = torch.tensor([dt], device=device, dtype=torch.float64)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    samples_per_step = T_STEPS * N_global * PPO_HP['ppo_epoch']
    value = samples_per_step * args.steps / dt

    # GAE scan timing (HBM-bound side kernel) on the same storage
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    ev0.record()
    for _ in range(20):
        rs.compute_returns(ro['next_value'], True, 0.99, 0.95, True)
    ev1.record()
    torch.cuda.synchronize()
    scan_ms = ev0.elapsed_time(ev1) / 20

    parity_ok = True
    if rank == 0:
        dom = max(prof.items(), key=lambda kv: kv[1]['total_ms']) if prof else (None, None)
        gemm_ms = sum(v['total_ms'] for v in prof.values())
        gemm_flop = sum(v['flops_per_call'] * v['calls'] for v in prof.values())
        roof = None
        traffic, traffic_src = _profile_traffic(dom[0], 'headline_kernel_stats', 'pmc_traffic')
        if dom[0] is not None:
            ach = dom[1]['flops_per_call'] / (dom[1]['avg_ms'] * 1e-3) / 1e12
            roof = dict(bound='mfma', kernel=dom[0], achieved=round(ach, 2), peak=MFMA_F32_PEAK_TFLOPS,
                        unit='TFLOP/s', frac=round(ach / MFMA_F32_PEAK_TFLOPS, 4), traffic=traffic, traffic_source=traffic_src,
                        avg_launch_ms=round(dom[1]['avg_ms'], 4), flop_per_launch=dom[1]['flops_per_call'])
        out = dict(metric='PPO update samples/sec (NatureCNN 84x84x4)', value=round(value, 1), unit='samples/s',
                   n_gpus=world, steps=args.steps, warmup=args.warmup, ms_per_step=round(dt / args.steps * 1e3, 3),
                   higher_is_better=True, scaling='weak', vs_baseline=None, dtype='f32', data='synthetic',
                   config=dict(workload='PPO.update NatureCNN 84x84x4 Categorical(6): T=128 x %d procs/GPU (%d global), '
                                        '4 epochs x 4 minibatches, fp32 MFMA' % (PROCS_PER_GPU, N_global),
                               global_batch=T_STEPS * N_global, minibatch=T_STEPS * N_global // PPO_HP['num_mini_batch'],
                               parallelism=dp_mode_string(world, rccl_world)),
                   roofline=roof,
                   whole_update=dict(mfma_tflops=round(samples_per_step * args.steps * FWD_BWD_FLOP_PER_SAMPLE / dt / 1e12 / world, 2),
                                     frac_of_f32_mfma_peak=round(samples_per_step * args.steps * FWD_BWD_FLOP_PER_SAMPLE / dt / 1e12 / world
                                                                 / MFMA_F32_PEAK_TFLOPS, 4),
                                     gemm_ops_ms_per_update=round(gemm_ms, 3),
                                     gemm_ops_tflops=round(gemm_flop / (gemm_ms * 1e-3) / 1e12, 2) if gemm_ms else None),
                   kernels={k: dict(calls=v['calls'], avg_ms=round(v['avg_ms'], 4),
                                    tflops=round(v['flops_per_call'] / (v['avg_ms'] * 1e-3) / 1e12, 2))
                            for k, v in sorted(prof.items(), key=lambda kv: -kv[1]['total_ms'])},
                   gae_scan=dict(T=T_STEPS, N=N_global, ms=round(scan_ms, 4),
                                 gbps=round(T_STEPS * N_global * 20 / (scan_ms * 1e-3) / 1e9, 2)),
                   losses=[round(x, 6) for x in losses])
        if world == 1 and not args.no_side_metrics:
            try:
                out['side_metrics'] = side_metrics(device)
                if 'roofline_kfac' in out['side_metrics']:
                    out['roofline_kfac'] = out['side_metrics'].pop('roofline_kfac')
            except Exception as e:                       # never lose the headline line over a side figure
                out['side_metrics'] = dict(error=repr(e))
        if world == 1 and not args.no_cpu_baseline:
            keep = {k: v.cpu() for k, v in ro.items()}
            cb, cpu_losses = cpu_baseline(keep, seed=7)
            cb['value'] = round(cb['value'], 1)
            out['cpu_baseline'] = cb
            # parity of what was timed: a FRESH agent (same initial weights, same CPU seed for the minibatch
            # permutations) runs the oracle's first update on the GPU; the three returned losses must agree to the
            # north_star bar (1e-4 relative)
            pol_c, agent_c = build_agent(device, seed=7)
            torch.manual_seed(7 + 1)
            gpu_losses = agent_c.update(rs)
            rel = [abs(a - b) / max(abs(b), 1e-3) for a, b in zip(gpu_losses, cpu_losses)]
            out['parity_check'] = dict(what='first PPO.update from identical weights / rollouts / CPU seed: GPU losses vs '
                                            'the cpu_baseline oracle run', gpu=[round(x, 7) for x in gpu_losses],
                                       cpu=[round(x, 7) for x in cpu_losses], max_rel_err=float('%.3g' % max(rel)),
                                       tol=1e-4, ok=bool(max(rel) <= 1e-4))
            parity_ok = out['parity_check']['ok']
            # the reference itself runs single-threaded (main.py:38 torch.set_num_threads(1)): one update that way too
            nthr = torch.get_num_threads()
            torch.set_num_threads(1)
            cb1, _ = cpu_baseline(keep, seed=7, max_updates=1)
            torch.set_num_threads(nthr)
            cb1['value'] = round(cb1['value'], 1)
            out['cpu_baseline_1thread'] = cb1
        sys.stdout.flush()
        os.write(json_fd, (json.dumps(out) + '\n').encode())
        if not parity_ok:
            sys.stderr.write('bench.py: GPU losses differ from the CPU oracle: %r\n' % (out['parity_check'],))
    if world > 1 or args.force_dp:
        dist.barrier()
        dist.destroy_process_group()
    if not parity_ok:
        sys.exit(3)


if __name__ == '__main__':
    main()
