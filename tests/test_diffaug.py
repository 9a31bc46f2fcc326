"""N2: DiffAugment / AugWrapper of the product against fixtures produced by the reference's own
stylex/diff_augment.py and AugWrapper.forward (oracle/make_golden.py::gen_diffaug).  The random parameters come from
the CPU generator in the reference's order, so outputs are reproduced exactly (fp32 arithmetic on the same values;
1e-6) on the CPU and on the GPU."""
import random

import numpy as np
import pytest
import torch

import diff_augment as da
import stylex_train as st
from conftest import load_golden


def seed_all(s):
    torch.manual_seed(s)
    np.random.seed(s)
    random.seed(s)


def check(device):
    g = load_golden("diffaug")
    x = torch.from_numpy(g["x"]).to(device)
    r = torch.from_numpy(g["r"]).to(device)
    assert sorted(da.AUGMENT_FNS) == [str(t) for t in g["types"]]
    for t in (str(t) for t in g["types"]):
        for rep in range(2):
            seed_all(int(g["%s/%d/seed" % (t, rep)]))
            xr = x.clone().requires_grad_()
            y = da.DiffAugment(xr, types=[t])
            (y * r).sum().backward()
            np.testing.assert_allclose(y.detach().cpu().numpy(), g["%s/%d/y" % (t, rep)], rtol=0, atol=1e-6, err_msg=t)
            np.testing.assert_allclose(xr.grad.cpu().numpy(), g["%s/%d/gx" % (t, rep)], rtol=0, atol=1e-6, err_msg=t)

    class Ident(torch.nn.Module):
        def forward(self, im):
            return im

    wrap = st.AugWrapper(Ident(), 16)
    flips = 0
    for k in range(6):
        seed_all(950 + k)
        y = wrap(x, prob=0.7, types=["translation", "cutout"], detach=True)
        np.testing.assert_allclose(y.cpu().numpy(), g["wrap/%d/y" % k], rtol=0, atol=1e-6)
        after = np.array([random.random(), float(torch.rand(()))])
        np.testing.assert_array_equal(after, g["wrap/%d/after" % k])  # consumed exactly the reference's draws
        flips += int(not np.allclose(g["wrap/%d/y" % k], g["x"]))
    assert flips >= 1  # the fixture exercises the augmented branch


def test_diffaugment_vs_reference_cpu():
    check(torch.device("cpu"))


@pytest.mark.gpu
def test_diffaugment_vs_reference_gpu():
    assert torch.cuda.is_available()
    check(torch.device("cuda:0"))
