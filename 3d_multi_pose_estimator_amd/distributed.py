"""Frame sharding across the GPUs of one node + one all-gather of the results.

Frames are independent units (the reference's loop body reads only its own frame,
test/metrics_from_model.py:120-300), so the path shards with no data-path collective: rank r
of G processes the contiguous frame range [r*ceil(B/G), ...).  The only exchange is one
all-gather per batch of the fixed-capacity result blocks (poses [frames, Pcap, J, 3] and
n_persons [frames]) — `torch.distributed` with backend "nccl" is RCCL over xGMI on ROCm; the
same code runs on "gloo" for the CPU tests.  Weights are replicated.
"""
import torch
import torch.distributed as dist


def shard_range(n_frames, rank, world):
    """Contiguous, near-equal shards; every rank gets the same capacity `per` (last ones padded)."""
    per = -(-n_frames // world)
    lo = min(n_frames, rank * per)
    hi = min(n_frames, lo + per)
    return lo, hi, per


def all_gather_results(poses, n_persons, n_frames_total, group=None):
    """poses [per, Pcap, J, 3], n_persons [per] of this rank (rows beyond the rank's real frame
    count must be zero) -> (poses [n_frames_total, ...], n_persons [n_frames_total]) on every rank.

    ONE collective per batch: the two blocks travel as one byte buffer per rank (the exchange is
    latency-bound -- 0.43 MB per rank at 5x4 -- so the second all-gather would cost as much as the
    first)."""
    world = dist.get_world_size(group)
    per = poses.shape[0]
    pb = poses.contiguous().view(torch.uint8).reshape(-1)
    nb = n_persons.contiguous().view(torch.uint8).reshape(-1)
    mine = torch.cat([pb, nb])
    out = torch.empty((world, mine.numel()), dtype=torch.uint8, device=poses.device)
    dist.all_gather_into_tensor(out.view(-1), mine, group=group)
    gp = out[:, :pb.numel()].contiguous().view(poses.dtype).reshape((world * per,) + tuple(poses.shape[1:]))
    gn = out[:, pb.numel():].contiguous().view(n_persons.dtype).reshape(world * per)
    return gp[:n_frames_total], gn[:n_frames_total]


def pad_to(t, per):
    if t.shape[0] == per:
        return t
    pad = torch.zeros((per - t.shape[0],) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
    return torch.cat([t, pad], dim=0)
