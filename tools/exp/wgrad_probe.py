import sys, os, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
import torch.nn.functional as F
from cp_pre_amd.convops_2d import ConvOperator
from cp_pre_amd.convops_1d import ConvOperator as Conv1D


def main():
    dev = torch.device("cuda:0")
    g = torch.Generator().manual_seed(3)
    for shape, kshape in (((3, 7, 19, 70), (3, 3, 3)), ((2, 5, 33, 300), (3, 3, 3)), ((4, 20, 130), (3, 3))):
        nd = len(kshape)
        x = torch.randn(*shape, generator=g)
        k = torch.randn(*kshape, generator=g)
        xd, kd = x.to(dev).requires_grad_(True), k.to(dev).requires_grad_(True)
        D = (ConvOperator if nd == 3 else Conv1D)()
        D.kernel = kd
        (D(xd) ** 2).sum().backward()
        xc, kc = x.clone().requires_grad_(True), k.clone().requires_grad_(True)
        conv = F.conv3d if nd == 3 else F.conv2d
        (conv(xc[:, None], kc[None, None], padding=1) ** 2).sum().backward()
        e1 = ((xd.grad.cpu() - xc.grad).abs().max() / xc.grad.abs().max()).item()
        e2 = ((kd.grad.cpu() - kc.grad).abs().max() / kc.grad.abs().max()).item()
        print(shape, kshape, "grad_field rel err %.2e  grad_kernel rel err %.2e" % (e1, e2))
    # timing: forward + backward with kernel grads
    x = torch.randn(64, 32, 128, 128, device=dev, requires_grad=True)
    D = ConvOperator()
    D.kernel = (ConvOperator('t', 2).kernel - 0.25 * ConvOperator(('x', 'y'), 2).kernel).to(dev).requires_grad_(True)
    def step():
        x.grad = None; D.kernel.grad = None
        (D(x) ** 2).sum().backward()
    for _ in range(2): step()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step()
    torch.cuda.synchronize(); print("fwd+bwd with kernel grad [64,32,128,128]: %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))
    kt = D.kernel.detach().clone().requires_grad_(True)
    xt = x.detach().clone().requires_grad_(True)
    def step_t():
        xt.grad = None; kt.grad = None
        (F.conv3d(xt[:, None], kt[None, None], padding=1) ** 2).sum().backward()
    for _ in range(2): step_t()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(5): step_t()
    torch.cuda.synchronize(); print("same through torch F.conv3d (MIOpen):       %.2f ms" % ((time.perf_counter() - t0) / 5 * 1e3))


if __name__ == "__main__":
    main()
