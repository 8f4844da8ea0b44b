"""Golden vectors for the GAIL discriminator / ExpertDataset from the REFERENCE (build container only).

    python tests/golden/make_golden_gail.py

Imports /root/reference/a2c_ppo_acktr/algo/gail.py through ref_harness (h5py stubbed; sb3's RunningMeanStd replaced
by the restatement in oracle/gail_ref.py -- that one call is parity-unpinned, see there) and records, for the cases
of tests/helpers.GAIL_CASES: two `Discriminator.update` calls (loss, parameters after each), three sequential
`predict_reward` calls, and an `ExpertDataset` built from a synthetic expert file.
"""
import os
import sys
import tempfile
from collections import OrderedDict

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))


class FixedLoader(object):
    """What gail.py:57-66 needs from the DataLoader: iteration over (state, action) CPU batches and .batch_size."""
    def __init__(self, batches, batch_size):
        self.batches, self.batch_size = batches, batch_size

    def __iter__(self):
        return iter(self.batches)


def main():
    import make_golden as mg          # installs the reference import harness
    import ref_harness
    from helpers import GAIL_CASES, gail_expert_file_tensors, gail_inputs, t
    from a2c_ppo_acktr.algo import gail          # reference
    out = {}
    for ci, (tag, od, A, hid, B, T, N) in enumerate(GAIL_CASES):
        P, ro, expert = gail_inputs(ci)
        disc = gail.Discriminator(od + A, hid, torch.device('cpu'))
        disc.load_state_dict(OrderedDict((k, t(v)) for k, v in P.items()), strict=True)
        rs = mg.fill_storage(ro, (od,), ref_harness.Box(A))
        loader = FixedLoader([(t(s), t(a)) for s, a in expert], B)
        for it in range(2):
            torch.manual_seed(900 + it)
            loss = disc.update(loader, rs, obsfilt=lambda x, update=False: x)
            out[tag + '/update%d/loss' % it] = np.float64(loss)
            for n, p in disc.named_parameters():
                out[tag + '/update%d/param/%s' % (it, n)] = p.detach().numpy().copy()
        # predict_reward: three consecutive steps of the rollout (main.py:150-155)
        for step in range(3):
            r = disc.predict_reward(t(ro['obs'][step]), t(ro['actions'][step]), 0.99, t(ro['masks'][step]))
            out[tag + '/reward%d' % step] = r.numpy().copy()
            out[tag + '/returns%d' % step] = disc.returns.numpy().copy()
            out[tag + '/rms_var%d' % step] = np.float64(disc.ret_rms.var[0])
        # ExpertDataset on a synthetic file in the gail_experts format
        with tempfile.TemporaryDirectory() as d:
            fn = os.path.join(d, 'trajs_synth.pt')
            torch.save(dict(gail_expert_file_tensors(ci)), fn)
            torch.manual_seed(950 + ci)
            ds = gail.ExpertDataset(fn, num_trajectories=4, subsample_frequency=20)
        out[tag + '/ds/len'] = np.int64(len(ds))
        out[tag + '/ds/get_idx'] = np.array(ds.get_idx, dtype=np.int64)
        for k in ('states', 'actions', 'lengths'):
            out[tag + '/ds/' + k] = ds.trajectories[k].numpy().copy()
        s0, a0 = ds[len(ds) - 1]
        out[tag + '/ds/last_state'] = s0.numpy().copy()
    np.savez_compressed(os.path.join(HERE, 'gail.npz'), **out)
    print('wrote gail.npz with', len(out), 'arrays')


if __name__ == '__main__':
    main()
