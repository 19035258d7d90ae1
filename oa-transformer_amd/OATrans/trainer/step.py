"""The hot path proper: ONE optimiser step (trainer_dist.py:156-168 of the reference), shared by the
trainers, bench.py and the smoke test so that what is benchmarked is what trains."""
try:
    from OATrans.model import layers as layers_mod
    from OATrans.model.layers import infonce_loss, sim_matrix
    from OATrans.model.loss import NormSoftmaxLoss
    from OATrans.model.oa_layers import bce_sum, mean_rows
    from OATrans.parallel import allgather_packed, allgather_pair
except ImportError:
    from model import layers as layers_mod
    from model.layers import infonce_loss, sim_matrix
    from model.loss import NormSoftmaxLoss
    from model.oa_layers import bce_sum, mean_rows
    from parallel import allgather_packed, allgather_pair


def _is_norm_softmax(loss_fn):
    """The library's NormSoftmaxLoss itself - whichever import path built it (`OATrans.model.loss` from the package, `model.loss`
    when an entry point runs from inside OATrans/) - and nothing else: a subclass may override forward, so it takes the general path."""
    cls = type(loss_fn)
    return cls is NormSoftmaxLoss or (cls.__qualname__ == "NormSoftmaxLoss" and cls.__module__.rsplit(".", 2)[-2:] == ["model", "loss"]
                                      and hasattr(loss_fn, "temperature"))


def _nce(loss_fn, t, v):
    """loss_fn(sim_matrix(t, v)); a NormSoftmaxLoss takes the one-node form (same kernels, no autograd glue in between)."""
    if layers_mod.HEAD_FUSED and _is_norm_softmax(loss_fn):
        return infonce_loss(t, v, loss_fn.temperature)
    return loss_fn(sim_matrix(t, v))


def hot_step(model_dp, loss_fn, optimizer, data, args):
    """forward -> packed all-gather -> sim_matrix -> loss -> backward -> grad all-reduce -> AdamW.
    Returns the loss as a device tensor (no host synchronisation)."""
    core = model_dp.module
    if hasattr(core, 'begin_step'):
        core.begin_step()
    optimizer.zero_grad()
    text_embeds, video_embeds = model_dp(data, aug=True)
    video_all, text_all = allgather_pair(video_embeds, text_embeds, args)
    loss = _nce(loss_fn, text_all, video_all)
    model_dp.backward(loss) if hasattr(model_dp, 'backward') else loss.backward()
    model_dp.sync_gradients()
    optimizer.step()
    return loss.detach()


def _finish(model_dp, optimizer, loss):
    model_dp.backward(loss) if hasattr(model_dp, 'backward') else loss.backward()
    model_dp.sync_gradients()
    optimizer.step()
    return loss.detach()


def region_mem_step(model_dp, loss_fn, optimizer, data, args):
    """trainer_region_mem.py:139-171: InfoNCE(text, video) + 0.1 * BCE_sum(region_sim, patch_mask) / rows,
    over the all-gathered batch."""
    core = model_dp.module
    core.begin_step()
    optimizer.zero_grad()
    text, video, rsim = model_dp(data, aug=True)
    patch_mask = data['patch_masks'].float()
    if patch_mask.dim() == 4:
        patch_mask = patch_mask.squeeze(1)
    video, text, rsim, patch_mask = allgather_packed([video, text, rsim, patch_mask], args)
    loss = _nce(loss_fn, text, video)
    rs = rsim.reshape(-1, rsim.shape[-1])
    pm = patch_mask.reshape(-1, patch_mask.shape[-1])
    loss = loss + 0.1 * bce_sum(rs, pm) / rs.shape[0]
    return _finish(model_dp, optimizer, loss)


def global_local_step(model_dp, loss_fn, optimizer, data, args):
    """trainer_global_local.py:144-215: NCE(text, video) + NCE(pad_text, video) + NCE(mean_o region, mean_o tags)."""
    core = model_dp.module
    core.begin_step()
    optimizer.zero_grad()
    text, pad_text, video, pad_video, extra = model_dp(data)
    region_feat, tags_feat = extra[4], extra[5]
    video, pad_text, pad_video, text, region_feat, tags_feat = allgather_packed(
        [video, pad_text, pad_video, text, region_feat, tags_feat], args)
    loss = _nce(loss_fn, text, video) + _nce(loss_fn, pad_text, video)
    loss = loss + _nce(loss_fn, mean_rows(region_feat), mean_rows(tags_feat))
    return _finish(model_dp, optimizer, loss)
