"""FlatAdam against torch.optim.Adam beyond the two reference steps of tests/golden/adam_steps.npz: a parameter that starts
receiving gradients LATER (a backbone unfrozen after N steps) must take its own step 1 - torch keeps state['step'] per parameter
(ADVICE r2); one parameter group only."""
import pytest
import torch

import s2lc_amd  # noqa: F401

pytestmark = pytest.mark.gpu
DEV = torch.device("cuda:0")


def test_unfrozen_parameters_start_their_own_bias_correction():
    from s2lc_amd.modules.efficientnet_unet import EfficientNetConfig, EfficientnetUnet
    from s2lc_amd.optim import FlatAdam

    torch.manual_seed(0)
    model = EfficientnetUnet(EfficientNetConfig("b0", 4, 4, class_distribution=[0.25] * 4)).to(DEV)
    ref = [p.detach().clone().requires_grad_(True) for p in model.parameters()]          # torch.optim.Adam on copies, same gradients
    opt = FlatAdam(model, lr=1e-2, weight_decay=0.05)
    topt = torch.optim.Adam(ref, lr=1e-2, weight_decay=0.05)
    names = [n for n, _ in model.named_parameters()]
    frozen = [n.startswith("encoder.") for n in names]
    g = torch.Generator(device=DEV).manual_seed(1)
    model._grad_buffer()
    model._publish_grads(model._no_grad_params)
    for step in range(7):
        unfreeze = step in (3, 4, 6)             # the encoder joins at the fourth step, is frozen AGAIN at the sixth and rejoins at the
                                                 # seventh: torch's state['step'] counts the steps a parameter was updated in
        for (n, p), r, fz in zip(model.named_parameters(), ref, frozen):
            if n in model._no_grad_params:
                p.grad, r.grad = None, None
                continue
            if fz and not unfreeze:
                p.grad, r.grad = None, None
                continue
            if p.grad is None:
                model._publish_grads(model._no_grad_params)
            p.grad.copy_(torch.randn(p.shape, device=DEV, generator=g) * 0.1)
            r.grad = p.grad.detach().clone()
        # _publish_grads re-attached every view: drop the frozen ones again for this step
        for (n, p), r, fz in zip(model.named_parameters(), ref, frozen):
            if fz and not unfreeze:
                p.grad, r.grad = None, None
        opt.step()
        topt.step()
    torch.cuda.synchronize()
    worst = 0.0
    for (n, p), r in zip(model.named_parameters(), ref):
        e = (p.detach() - r.detach()).abs().max().item() / max(r.detach().abs().max().item(), 1e-12)
        worst = max(worst, e)
        assert e < 2e-6, (n, e)
    print(f"FlatAdam vs torch.optim.Adam after freeze -> unfreeze: worst relative difference {worst:.2e}")
    st = opt.state_dict()
    assert st["state"]["step"] == 7 and set(st["state"]["updates"].values()) == {7, 3}         # two cohorts of parameters: 7 and 3 updates
    opt2 = FlatAdam(model, lr=1e-2, weight_decay=0.05)
    opt2.load_state_dict(st)
    assert opt2._updates == opt._updates and opt2.step_count == 7
    with pytest.raises(ValueError, match="one parameter group"):
        opt.add_param_group({"params": [torch.nn.Parameter(torch.zeros(1, device=DEV))]})
