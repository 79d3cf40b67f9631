import sys, os, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from cp_pre_amd.convops_1d import ConvOperator as C1
from cp_pre_amd.convops_2d import ConvOperator as C2
import torch.nn.functional as F
g = torch.Generator().manual_seed(1)
for shape in [(2, 13, 7), (2, 13, 8), (2, 13, 4), (2, 13, 5), (3, 9, 260), (1, 40, 12)]:
    x = torch.randn(*shape, generator=g)
    ker = torch.zeros(3, 3); ker[0, 1] = 1.5; ker[2, 1] = -0.5; ker[1, 0] = 2.0; ker[1, 2] = -3.0; ker[1, 1] = 0.25
    D = C1(); D.kernel = ker
    got = D(x.cuda()).cpu()
    want = F.conv2d(x[:, None], ker[None, None], padding=1)[:, 0]
    err = (got - want).abs()
    print(shape, "max err", float(err.max()), "bad cols", sorted(set(torch.nonzero(err > 1e-4)[:, 2].tolist()))[:10],
          "bad rows", sorted(set(torch.nonzero(err > 1e-4)[:, 1].tolist()))[:10])
for shape in [(2, 5, 13, 260), (1, 3, 9, 516)]:
    x = torch.randn(*shape, generator=g)
    D = C2(("x", "y"), 2); D.kernel = D.kernel + 0.5 * C2("t", 2).kernel
    got = D(x.cuda()).cpu()
    want = F.conv3d(x[:, None], D.kernel[None, None], padding=1)[:, 0]
    err = (got - want).abs()
    print(shape, "max err", float(err.max()), "bad cols", sorted(set(torch.nonzero(err > 1e-4)[:, 3].tolist()))[:10])
