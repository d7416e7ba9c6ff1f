"""CPU restatement of the reference's densification semantics.   *** TEST INFRASTRUCTURE ONLY ***

Follows reference scene/gaussian_model.py:367-429 (densify_and_split, densify_and_clone, densify_and_prune) and the optimizer
surgery of :274-364 step by step on plain tensors (the reference file itself cannot be imported: SyntaxError at :410-412,
SURVEY.md 0.3).  The split's `torch.normal` samples are taken as an argument so tests can compare the deterministic structure
exactly and the random part statistically.  Parity unpinned (the reference holds no fixture for it); pinned by construction
against the cited lines.
"""
import torch


def build_rotation(r):
    """reference utils/general_utils.py:78-99"""
    q = r / r.norm(dim=1, keepdim=True)
    w, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    R = torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - w * z), 2 * (x * z + w * y),
                     2 * (x * y + w * z), 1 - 2 * (x * x + z * z), 2 * (y * z - w * x),
                     2 * (x * z - w * y), 2 * (y * z + w * x), 1 - 2 * (x * x + y * y)], dim=1)
    return R.view(-1, 3, 3)


def densify_and_prune(params, moments, xyz_gradient_accum, denom, max_radii2D, max_grad, min_opacity, extent,
                      max_screen_size, percent_dense, normal_samples=None, N=2):
    """params: dict name -> tensor (xyz, f_dc, f_rest, opacity, scaling, rotation); moments: dict name -> (exp_avg, exp_avg_sq) or
    None.  Returns (new_params, new_moments, info) with info['source'] = source row of every output row, info['kind'] in
    {0 kept, 1 clone, 2 child}."""
    p = {k: v.clone() for k, v in params.items()}
    m = {k: (None if v is None else (v[0].clone(), v[1].clone())) for k, v in moments.items()}
    n0 = p["xyz"].shape[0]
    source = torch.arange(n0)
    kind = torch.zeros(n0, dtype=torch.long)

    def cat(new, src, kd):                                       # cat_tensors_to_optimizer + densification_postfix
        nonlocal source, kind, max_radii2D
        for k in p:
            p[k] = torch.cat((p[k], new[k]), dim=0)
            if m[k] is not None:
                m[k] = (torch.cat((m[k][0], torch.zeros_like(new[k])), 0), torch.cat((m[k][1], torch.zeros_like(new[k])), 0))
        source = torch.cat((source, src))
        kind = torch.cat((kind, torch.full_like(src, kd)))
        max_radii2D = torch.zeros(p["xyz"].shape[0])            # :364

    def prune(mask):                                             # prune_points / _prune_optimizer
        nonlocal source, kind, max_radii2D
        keep = ~mask
        for k in p:
            p[k] = p[k][keep]
            if m[k] is not None:
                m[k] = (m[k][0][keep], m[k][1][keep])
        source, kind, max_radii2D = source[keep], kind[keep], max_radii2D[keep]

    grads = xyz_gradient_accum / denom                           # :414-415
    grads[grads.isnan()] = 0.0
    get_scaling = lambda: torch.exp(p["scaling"])                # noqa: E731
    # ---- clone (:389-408)
    sel = torch.where(torch.norm(grads, dim=-1) >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(get_scaling(), dim=1).values <= percent_dense * extent)
    cat({k: v[sel] for k, v in p.items()}, torch.nonzero(sel).flatten(), 1)
    # ---- split (:367-387)
    n_init = p["xyz"].shape[0]
    padded = torch.zeros(n_init)
    padded[:grads.shape[0]] = grads.squeeze(-1)
    sel = torch.where(padded >= max_grad, True, False)
    sel = torch.logical_and(sel, torch.max(get_scaling(), dim=1).values > percent_dense * extent)
    stds = get_scaling()[sel].repeat(N, 1)
    z = normal_samples if normal_samples is not None else torch.randn(stds.shape[0], 3)
    samples = stds * z
    rots = build_rotation(p["rotation"][sel]).repeat(N, 1, 1)
    new = {
        "xyz": torch.bmm(rots, samples.unsqueeze(-1)).squeeze(-1) + p["xyz"][sel].repeat(N, 1),
        "scaling": torch.log(get_scaling()[sel].repeat(N, 1) / (0.8 * N)),
        "rotation": p["rotation"][sel].repeat(N, 1),
        "f_dc": p["f_dc"][sel].repeat(N, 1, 1),
        "f_rest": p["f_rest"][sel].repeat(N, 1, 1),
        "opacity": p["opacity"][sel].repeat(N, 1),
    }
    src = torch.nonzero(sel).flatten().repeat(N)
    cat(new, src, 2)
    prune(torch.cat((sel, torch.zeros(N * int(sel.sum()), dtype=torch.bool))))
    # ---- prune (:418-424)
    prune_mask = (torch.sigmoid(p["opacity"]) < min_opacity).squeeze(-1)
    if max_screen_size:
        big_vs = max_radii2D > max_screen_size
        big_ws = get_scaling().max(dim=1).values > 0.1 * extent
        prune_mask = torch.logical_or(torch.logical_or(prune_mask, big_vs), big_ws)
    prune(prune_mask)
    return p, m, dict(source=source, kind=kind)
