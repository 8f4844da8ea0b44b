"""CPU: the whole-batch expert fetch used by Discriminator.update (a2c_ppo_acktr/algo/gail.py `_expert_batches`) returns
the rows a plain iteration of the same DataLoader returns (reference gail.py:57-66 iterates the loader), in the same
order, and leaves the CPU generator where the plain iteration leaves it; other loaders are passed through."""
import os
import tempfile

import torch
import torch.utils.data


def _dataset(seed):
    from a2c_ppo_acktr.algo import gail
    g = torch.Generator().manual_seed(seed)
    with tempfile.TemporaryDirectory() as d:
        fn = os.path.join(d, 'trajs.pt')
        torch.save({'states': torch.randn(7, 200, 5, generator=g), 'actions': torch.randn(7, 200, 2, generator=g),
                    'rewards': torch.randn(7, 200, generator=g), 'lengths': torch.full((7,), 200, dtype=torch.int64)}, fn)
        torch.manual_seed(seed)
        return gail.ExpertDataset(fn, num_trajectories=4, subsample_frequency=20)


def test_rows_match_getitem():
    ds = _dataset(3)
    index = [0, len(ds) - 1, 5, 5, 17]
    want = torch.utils.data.default_collate([ds[i] for i in index])
    got = ds.rows(index)
    assert torch.equal(got[0], want[0]) and torch.equal(got[1], want[1])


def test_fast_batches_equal_loader_batches_and_rng_state():
    from a2c_ppo_acktr.algo import gail
    ds = _dataset(4)
    for shuffle, drop_last in ((True, True), (True, False), (False, False)):
        loader = torch.utils.data.DataLoader(ds, batch_size=16, shuffle=shuffle, drop_last=drop_last)
        torch.manual_seed(77)
        plain = [b for b in loader]
        tail_plain = torch.rand(3)
        torch.manual_seed(77)
        fast = [b for b in gail._expert_batches(loader)]
        tail_fast = torch.rand(3)
        assert len(plain) == len(fast) == len(loader)
        for a, b in zip(plain, fast):
            assert torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])
        assert torch.equal(tail_plain, tail_fast), 'the CPU generator must be left in the same state'


def test_other_loaders_pass_through():
    from a2c_ppo_acktr.algo import gail
    ds = _dataset(5)
    loader = torch.utils.data.DataLoader(ds, batch_size=8, shuffle=False, collate_fn=lambda items: len(items))
    assert list(gail._expert_batches(loader)) == [8] * (len(ds) // 8) + ([len(ds) % 8] if len(ds) % 8 else [])
    assert list(gail._expert_batches([1, 2, 3])) == [1, 2, 3]
