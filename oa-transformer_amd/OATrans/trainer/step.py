"""The hot path proper: ONE optimiser step (trainer_dist.py:156-168 of the reference), shared by the
trainers, bench.py and the smoke test so that what is benchmarked is what trains."""
try:
    from OATrans.model.layers import sim_matrix
    from OATrans.parallel import allgather_pair
except ImportError:
    from model.layers import sim_matrix
    from parallel import allgather_pair


def hot_step(model_dp, loss_fn, optimizer, data, args):
    """forward -> packed all-gather -> sim_matrix -> loss -> backward -> grad all-reduce -> AdamW.
    Returns the loss as a device tensor (no host synchronisation)."""
    core = model_dp.module
    if hasattr(core, 'begin_step'):
        core.begin_step()
    optimizer.zero_grad()
    text_embeds, video_embeds = model_dp(data, aug=True)
    video_all, text_all = allgather_pair(video_embeds, text_embeds, args)
    loss = loss_fn(sim_matrix(text_all, video_all))
    loss.backward()
    model_dp.sync_gradients()
    optimizer.step()
    return loss.detach()
