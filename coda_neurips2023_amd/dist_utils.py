"""Cross-rank helpers the reference's engine.py calls around the path (utils/dist.py:67-115, 164-185).

``all_gather_dict`` keeps the reference's contract -- every tensor of a dictionary concatenated over the ranks along
dim 0, ``logit_scale`` and non-tensors dropped when distributed -- with one ``all_gather_into_tensor`` per entry
into a preallocated output (RCCL writes the concatenation directly; the reference allocates ``world`` temporaries
and concatenates).  ``skip`` names entries that are not worth moving: engine.py:2625 gathers the whole batch
dictionary, i.e. every rank receives every other rank's 20 000-point clouds (1.9 MB per rank and step) only for the
empty-box test -- ``APCalculator.step_meter`` here filters on the local rank and ``merge_across_ranks`` exchanges the
few surviving rows once, at the end."""
import torch
import torch.distributed as dist


def is_distributed():
    return dist.is_available() and dist.is_initialized()


def get_world_size():
    return dist.get_world_size() if is_distributed() else 1


def get_rank():
    return dist.get_rank() if is_distributed() else 0


def all_reduce_average(tensor):
    """utils/dist.py:67-87: the loss summed over the ranks / world size; the tensor itself when not distributed.
    (engine.py:152 calls it on the loss every training step.)"""
    if not is_distributed():
        return tensor
    t = tensor.detach()
    t = t[None] if t.ndim == 0 else t.clone()
    dist.all_reduce(t)
    return (t.squeeze(0) if tensor.ndim == 0 else t) / get_world_size()


def reduce_dict(input_dict, average=True):
    """utils/dist.py:91-115 (engine.py:153): every value of the loss dictionary averaged over the ranks -- keys
    sorted so that all ranks stack them in the same order, ONE all-reduce of the stacked scalars."""
    world = get_world_size()
    if world < 2:
        return input_dict
    with torch.no_grad():
        names = sorted(input_dict.keys())
        values = torch.stack([input_dict[k].detach().reshape(()) for k in names], dim=0)
        dist.all_reduce(values)
        if average:
            values /= world
        return dict(zip(names, values))


def all_gather_dict(data, skip=()):
    assert isinstance(data, dict)
    out = {}
    world = get_world_size()
    for key, value in data.items():
        if key in skip:
            continue
        if isinstance(value, torch.Tensor) and key != "logit_scale":
            if is_distributed():
                value = value.contiguous()
                gathered = torch.empty((world * value.shape[0],) + tuple(value.shape[1:]), dtype=value.dtype,
                                       device=value.device)
                if value.is_cuda:
                    dist.all_gather_into_tensor(gathered, value)
                else:  # gloo: no single-tensor form
                    parts = list(gathered.chunk(world, 0))
                    dist.all_gather(parts, value)
                out[key] = gathered
            else:
                out[key] = value
        elif not is_distributed():
            out[key] = value
    return out
