"""helpers shared by the conditional-Glow tests (no reference code: the perturbation below restates, for the TEST's
purposes, the recipe tools/gen_golden.py applied to the reference's freshly constructed net before recording G19)"""
import torch


def perturb_glow(net, gen, amp=1.0):
    with torch.no_grad():
        for k, p in net.named_parameters():
            r = (amp * torch.randn(p.shape, generator=gen)).to(p.device)
            if k.endswith('.scale'):
                p.add_(0.1 * r)
            elif 'conv_zero' in k or 'top_latent' in k or 'latent_encoder' in k:
                p.add_((0.02 if k.endswith('weight') else 0.1) * r)
            elif k.endswith('.norm.weight'):
                p.copy_(1 + 0.1 * r)
            elif 'norm' in k and k.endswith('.weight'):
                p.copy_(1 + 0.2 * r)
            elif k.endswith('.bias'):
                p.add_(0.1 * r)
            elif k.endswith('.l') or k.endswith('.u'):
                p.add_(0.05 * r)
            elif k.endswith('.log_s'):
                p.add_(0.1 * r)
            elif k.endswith('conv1x1.weight'):
                p.add_(0.05 * r)


def reverse_kl(net, x, eps, beta, weight_bound):
    """train_cglow_reverse_kl.py:250-262 on the drop-in modules -> (loss, loss_pde, neg_entropy, y, logp)"""
    import math
    from pde_surrogate_amd.models.darcy import darcy_mixed_residual_loss
    y, logp = net.generate(x, eps_list=eps)
    loss_pde = darcy_mixed_residual_loss(x, y, weight_bound)[0]
    neg_entropy = logp.mean() / math.log(2.) / y[0].numel()
    return loss_pde * beta + neg_entropy, loss_pde, neg_entropy, y, logp
