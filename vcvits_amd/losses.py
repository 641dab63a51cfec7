"""vits/light/losses.py of the reference: LSGAN generator/discriminator losses, feature matching,
masked KL -- each one fused reduction node over all discriminator outputs (no per-tensor host
sync: the reference's `.item()` logging lists are returned as device tensors)."""
from . import ops


def feature_loss(fmap_r, fmap_g):
    """losses.py:4-12: 2 * sum over all feature maps of mean|r - g| (r detached)."""
    rs = [rl.detach() for dr in fmap_r for rl in dr]
    gs = [gl for dg in fmap_g for gl in dg]
    return ops.l1_mean_sum(gs, rs, weight=2.0)


def discriminator_loss(disc_real_outputs, disc_generated_outputs):
    """losses.py:14-27.  Returns (loss, r_losses, g_losses); the per-discriminator terms are
    0-dim device tensors instead of Python floats (no .item() sync)."""
    r_terms = ops.sq_mean_terms(list(disc_real_outputs), 1.0)
    g_terms = ops.sq_mean_terms(list(disc_generated_outputs), 0.0)
    loss = r_terms.sum() + g_terms.sum()
    return loss, list(r_terms.detach().unbind(0)), list(g_terms.detach().unbind(0))


def generator_loss(disc_outputs):
    """losses.py:29-38"""
    terms = ops.sq_mean_terms(list(disc_outputs), 1.0)
    return terms.sum(), list(terms.unbind(0))


def kl_loss(z_p, logs_q, m_p, logs_p, z_mask):
    """losses.py:40-55"""
    return ops.kl_loss(z_p, logs_q, m_p, logs_p, z_mask)
