"""GPU: la_adamw_step / FlatAdamW against torch.optim.AdamW (the reference's optimizer) over several steps, with warm-up."""
import pytest
import torch

from labelanything_amd.optim import FlatAdamW, constant_with_warmup

pytestmark = pytest.mark.gpu


def test_flat_adamw_tracks_torch_adamw_with_warmup():
    from transformers import get_scheduler
    g = torch.Generator().manual_seed(0)
    shapes = [(257, 33), (1000,), (4, 5, 6), (1,)]
    ref_params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    dev_params = [p.detach().clone().cuda() for p in ref_params]
    ref_opt = torch.optim.AdamW(ref_params, lr=5e-5)
    sched = get_scheduler("constant_with_warmup", optimizer=ref_opt, num_warmup_steps=3, num_training_steps=100)
    opt = FlatAdamW(dev_params, lr=5e-5, num_warmup_steps=3)
    assert all(p.data_ptr() == v.data_ptr() for p, v in zip(dev_params, opt.views))       # parameters are views of the flat buffer
    for step in range(8):
        lr_ref = sched.get_last_lr()[0]
        assert abs(opt.lr - lr_ref) < 1e-12
        for p, gv in zip(ref_params, opt.grad_views):
            grad = torch.randn(p.shape, generator=g) * (10.0 if step == 2 else 1.0)
            p.grad = grad.clone()
            gv.copy_(grad)
        ref_opt.step()
        sched.step()
        opt.step()
        torch.cuda.synchronize()
        for p, d in zip(ref_params, dev_params):
            diff = float((d.cpu() - p.detach()).abs().max())
            assert diff <= 2e-7 * max(1.0, float(p.detach().abs().max())), (step, diff)
    assert constant_with_warmup(0, 3) == 0.0 and constant_with_warmup(3, 3) == 1.0


def test_flat_adamw_rejects_cpu_parameters():
    with pytest.raises(RuntimeError, match="device parameters"):
        FlatAdamW([torch.zeros(3)])


def test_flat_adamw_parameter_groups_like_backbone_lr():
    """The reference's ``backbone_lr`` branch (models/lam.py:340-346): two torch parameter groups with their own rates under one
    warm-up schedule == FlatAdamW(lrs=...)."""
    from transformers import get_scheduler
    g = torch.Generator().manual_seed(1)
    shapes = [(64, 16), (16,), (300,), (7, 9)]
    ref_params = [torch.nn.Parameter(torch.randn(*s, generator=g)) for s in shapes]
    dev_params = [p.detach().clone().cuda() for p in ref_params]
    ref_opt = torch.optim.AdamW([{"params": ref_params[:2], "lr": 1e-5}, {"params": ref_params[2:]}], lr=5e-5)
    sched = get_scheduler("constant_with_warmup", optimizer=ref_opt, num_warmup_steps=2, num_training_steps=100)
    opt = FlatAdamW(dev_params, lr=5e-5, num_warmup_steps=2, lrs=[1e-5, 1e-5, 5e-5, 5e-5])
    for step in range(5):
        for p, gv in zip(ref_params, opt.grad_views):
            grad = torch.randn(p.shape, generator=g)
            p.grad = grad.clone()
            gv.copy_(grad)
        ref_opt.step()
        sched.step()
        opt.step()
        torch.cuda.synchronize()
        for p, d in zip(ref_params, dev_params):
            assert float((d.cpu() - p.detach()).abs().max()) <= 2e-7 * max(1.0, float(p.detach().abs().max())), step
