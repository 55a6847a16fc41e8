"""Multi-GPU sharding of a verification batch: contiguous item ranges per rank, no data-path collective, and one
all-gather of the packed accept bitmap (RCCL over xGMI on GPUs -- torch.distributed backend "nccl" -- or gloo in the
CPU tests). One process per GPU."""
import torch
import torch.distributed as dist


def shard_range(n_total, rank, world):
    """Rank `rank` verifies items [lo, hi) -- contiguous block partition."""
    lo = n_total * rank // world
    hi = n_total * (rank + 1) // world
    return lo, hi


def bitmap_words(n_items):
    return (n_items + 63) // 64


def pack_bits(results):
    """uint8 0/1 tensor -> int64 words, bit (i % 64) of word i // 64 = results[i] (same layout as k_pack)."""
    n = results.numel()
    w = bitmap_words(n)
    padded = torch.zeros(w * 64, dtype=torch.int64, device=results.device)
    padded[:n] = results.to(torch.int64)
    shifts = torch.arange(64, dtype=torch.int64, device=results.device)
    # bit 63 lands in the sign bit of int64; summing disjoint powers of two is exact in two's complement
    return (padded.view(w, 64) << shifts).sum(dim=1)


def unpack_bits(words, n_items):
    shifts = torch.arange(64, dtype=torch.int64, device=words.device)
    return ((words.view(-1, 1) >> shifts) & 1).reshape(-1)[:n_items].to(torch.uint8)


def all_gather_bitmap(local_words, world=None, out=None):
    """Every rank contributes its shard's bitmap words (equal length on all ranks); returns the concatenation in
    rank order. The only collective of the pipeline."""
    if world is None:
        world = dist.get_world_size() if dist.is_initialized() else 1
    if world == 1:
        return local_words
    if out is None:
        out = torch.empty(local_words.numel() * world, dtype=local_words.dtype, device=local_words.device)
    if local_words.is_cuda and dist.get_backend() == "gloo":
        # several ranks sharing ONE GPU (bench.py's MBLS_BENCH_SHARE_GPU test mode: RCCL wants one rank per device): gloo gathers host tensors
        host = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(host, local_words.cpu())
        out.copy_(host)
        return out
    dist.all_gather_into_tensor(out, local_words)
    return out


def all_gather_records(mine, world=None, out=None):
    """verify_multiple cut into shards (include/mbls.h, mbls_verify_multiple_partial_device): every rank contributes its one
    MBLS_VM_PARTIAL_BYTES record (a uint8 tensor); returns the records of all ranks in rank order -- the order every rank hands to
    mbls_verify_multiple_finish_device. The one exchange step of that path."""
    return all_gather_bitmap(mine, world, out)

