"""Same-box A/B of library builds: PPO.update (headline shape) timed in alternating child processes, one per library.

    python tools/ab_update.py libA.so libB.so [...] [--rounds 3] [--updates 30]

Each child loads ONE library through ACKTR_HIP_LIB, warms up, times `updates` updates (median ms) and prints per-operator
HIP-event times; the parent alternates A, B, ..., A, B, ... so that clock / temperature drift hits every build alike.
"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n_upd):
    sys.path[:0] = [ROOT, os.path.join(ROOT, 'pytorch-a2c-ppo-acktr-gail_amd')]
    import statistics
    import torch
    import bench
    from a2c_ppo_acktr import ops
    dev = torch.device('cuda:0')
    ro = bench.synth_rollouts(bench.T_STEPS, bench.PROCS_PER_GPU, dev, seed=1234)
    if os.environ.get('AB_U8'):
        ro['obs'] = ro['obs'].to(torch.uint8)
    pol, agent = bench.build_agent(dev, seed=7)
    rs = bench.fill_storage(ro, dev)
    rs.compute_returns(ro['next_value'], True, 0.99, 0.95, True)
    bench._timed_updates(agent, rs, 5, 100)
    ms = bench._timed_updates(agent, rs, n_upd, 200)
    for _ in range(2):
        ops.profiler = {}
        agent.update(rs)
    prof = {k: round(v['avg_ms'] * 1e3, 1) for k, v in ops.profile_summary().items()}
    ops.profiler = None
    print(json.dumps(dict(ms=round(statistics.median(ms), 4), ops=prof)))


def main():
    if sys.argv[1] == '--child':
        return child(int(sys.argv[2]))
    libs = [a for a in sys.argv[1:] if a.endswith('.so') or ('=' in a and not a.startswith('--'))]   # a library, or VAR=VALUE on the in-tree one
    rounds = int(sys.argv[sys.argv.index('--rounds') + 1]) if '--rounds' in sys.argv else 3
    upd = int(sys.argv[sys.argv.index('--updates') + 1]) if '--updates' in sys.argv else 30
    res = {l: [] for l in libs}
    for r in range(rounds):
        for l in libs:
            env = dict(os.environ)
            if l.endswith('.so'):
                env['ACKTR_HIP_LIB'] = os.path.abspath(l)
            else:
                env[l.split('=', 1)[0]] = l.split('=', 1)[1]
            out = subprocess.run([sys.executable, os.path.abspath(__file__), '--child', str(upd)], env=env,
                                 capture_output=True, text=True, timeout=300)
            if out.returncode != 0:
                print(l, 'FAILED', out.stderr[-800:], flush=True)
                return 1
            rec = json.loads(out.stdout.strip().splitlines()[-1])
            res[l].append(rec)
            print('%-40s round %d  %.3f ms' % (os.path.basename(l), r, rec['ms']), flush=True)
    for l in libs:
        ms = sorted(x['ms'] for x in res[l])
        print('%-40s median %.3f  min %.3f' % (os.path.basename(l), ms[len(ms) // 2], ms[0]))
        keys = res[l][0]['ops'].keys()
        print('    ' + '  '.join('%s %.1f' % (k.replace('linear1568x512', 'fc'), min(x['ops'][k] for x in res[l])) for k in keys))
    return 0


if __name__ == '__main__':
    sys.exit(main())
