"""Diagnostic: count ReLU gates of the cnn80 golden case whose sign differs between the HIP forward and an fp64 forward."""
import os, sys
import numpy as np, torch, torch.nn.functional as F
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in ('pytorch-a2c-ppo-acktr-gail_amd', 'tests', 'tests/golden', ''):
    sys.path.insert(0, os.path.join(ROOT, p))
import synth
from helpers import EVAL_CASES, eval_case_inputs, shapes_for, t
from a2c_ppo_acktr import ops
ci = 3
tag, kind, od, A, B, clip, c_e = EVAL_CASES[ci]
P = synth.make_params(shapes_for(kind, od, A), 40 + ci)
x = t(eval_case_inputs(tag, kind, od, A, B, ci)['obs']).cuda()
names = [('base.main.0', 4), ('base.main.2', 2), ('base.main.4', 1)]
a, a64 = x, x.double() / 255.0
for li, (n, s) in enumerate(names):
    w, b = t(P[n + '.weight']).cuda(), t(P[n + '.bias']).cuda()
    y = ops.conv2d_fwd(a, w, b, s, in_scale=(1 / 255.0 if li == 0 else 1.0), relu=True)
    pre64 = F.conv2d(a64, w.double(), b.double(), stride=s)
    flips = ((y > 0) != (pre64 > 0))
    print(n, 'flips', int(flips.sum()), 'of', y.numel(), 'min |pre| at flips', pre64[flips].abs().max().item() if flips.any() else None,
          'near-zero count(|pre|<1e-6):', int((pre64.abs() < 1e-6).sum()))
    a, a64 = y, torch.relu(pre64)
