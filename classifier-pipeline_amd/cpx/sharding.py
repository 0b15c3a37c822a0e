"""Multi-GPU layout of the path: clips are independent, so they shard across
ranks with no data-path collective (reference parallelism: multiprocessing.Pool
over files, track/trackextractor.py:80-85).  The only exchange is one
all-gather of fixed-width per-clip result records so that every rank (or
rank 0) can write the metadata."""




def partition_clips(frame_counts, world_size):
    """Greedy longest-processing-time partition of clips by frame count.
    -> list (per rank) of clip indices, each ascending; deterministic."""
    order = sorted(range(len(frame_counts)), key=lambda i: (-int(frame_counts[i]), i))
    loads = [0] * world_size
    shards = [[] for _ in range(world_size)]
    for i in order:
        r = min(range(world_size), key=lambda k: (loads[k], k))
        shards[r].append(i)
        loads[r] += int(frame_counts[i])
    return [sorted(s) for s in shards]


def rank_world():
    """(rank, world_size, local_rank) of this process under torchrun / torch.distributed.run, (0, 1, 0) otherwise."""
    import os

    try:
        return (int(os.environ.get("RANK", 0)), max(1, int(os.environ.get("WORLD_SIZE", 1))),
                int(os.environ.get("LOCAL_RANK", 0)))
    except ValueError:
        return 0, 1, 0


def usable_cpus():
    """CPUs this process can actually run on: the scheduler affinity, capped by the cgroup v2 / v1 CPU quota (a GPU box
    shows 256 logical CPUs and grants 16)."""
    import os

    try:
        n = len(os.sched_getaffinity(0))
    except AttributeError:
        n = os.cpu_count() or 1
    try:
        with open("/sys/fs/cgroup/cpu.max") as fh:
            quota, period = fh.read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except (OSError, ValueError):
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as fh:
                q = int(fh.read())
            with open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as fh:
                per = int(fh.read())
            if q > 0:
                n = min(n, max(1, q // per))
        except (OSError, ValueError):
            pass
    return max(1, n)


def host_threads_per_rank(cap=16):
    """Worker threads one rank's host stages may start (file reads into pinned memory): the usable CPUs shared among
    the ranks of this node (LOCAL_WORLD_SIZE / WORLD_SIZE under torchrun), at most `cap`.  Eight ranks that each
    start min(16, cpu_count()) readers beside their three pipeline threads oversubscribe a 16-CPU quota eight times
    (VERDICT r03 item 5c); the reference's counterpart is one worker process per file (trackextractor.py:80-85)."""
    import os

    try:
        local_world = int(os.environ.get("LOCAL_WORLD_SIZE") or os.environ.get("WORLD_SIZE") or 1)
    except ValueError:
        local_world = 1
    return max(1, min(cap, usable_cpus() // max(1, local_world)))


def shard_files(paths, rank, world_size):
    """This rank's share of a list of recordings: longest-processing-time partition by file size (a CPTV file's size
    tracks its frame count), identical on every rank; order of `paths` is kept inside a shard."""
    import os

    if world_size <= 1:
        return list(paths)
    sizes = [os.path.getsize(p) for p in paths]
    return [paths[i] for i in partition_clips(sizes, world_size)[rank]]


class GatheredRecords:
    """What the step's one collective returned: every rank's slab `[1 + capacity, width]` side by side.  Row 0 of a slab
    is its header (column 0: the rank's true record count, the rest 0), the records follow, unused rows are -1.  Nothing
    here reads the device: `slabs` and `counts` are device tensors a later kernel (or the next step) can consume;
    `records()` is the one place that synchronises, for the caller that writes metadata on the host."""

    def __init__(self, slabs, capacity, world):
        self.slabs = slabs              # [world, 1 + capacity, width]
        self.capacity = int(capacity)
        self.world = int(world)

    @property
    def width(self):
        return int(self.slabs.shape[2])

    @property
    def counts(self):
        """Per-rank true record counts, on the slabs' device (a view: no copy, no sync)."""
        return self.slabs[:, 0, 0]

    def records(self):
        """-> [n, width], the padding dropped, sorted by clip id (stable: a rank's own order is kept inside a clip);
        identical on every rank.  Raises on every rank alike when some rank had more records than the capacity the plan
        promised (its surplus never entered the collective)."""
        body = self.slabs[:, 1:, :].reshape(-1, self.width)
        counts = self.counts
        if bool((counts > self.capacity).any()):
            raise RuntimeError("gather_records: a rank produced %d records, capacity %d: size the capacity from the "
                               "plan (clips per rank x tracks per clip)" % (int(counts.max()), self.capacity))
        import torch

        out = body[body[:, 0] >= 0]
        return out[torch.argsort(out[:, 0], stable=True)]


def gather_records(records, dist=None, capacity=None):
    """records: int32 tensor [n_local, width] (clip id, >= 0, in column 0) -> GatheredRecords.
    ONE collective per step and no host synchronisation (what replaces the reference's
    multiprocessing.Pool over files, track/trackextractor.py:80-85): every rank contributes a slab of the same,
    plan-derived size -- `capacity` rows, e.g. clips per rank x tracks per clip, known before the step runs -- with its
    record count carried in the slab's header row, so no rank has to learn another's count before the exchange.
    `capacity` may be omitted only outside a process group (one rank: the slab is the records themselves)."""
    import torch

    n, width = int(records.shape[0]), int(records.shape[1])
    grouped = dist is not None and dist.is_initialized()
    if capacity is None:
        if grouped:
            raise ValueError("gather_records under a process group needs the plan's capacity (rows per rank)")
        capacity = n
    capacity = int(capacity)
    slab = torch.full((1 + capacity, width), -1, dtype=records.dtype, device=records.device)
    slab[0] = 0
    slab[0, 0] = n
    keep = min(n, capacity)
    slab[1:1 + keep] = records[:keep]
    if not grouped:
        return GatheredRecords(slab.unsqueeze(0), capacity, 1)
    world = dist.get_world_size()
    slabs = torch.empty((world * (1 + capacity), width), dtype=records.dtype, device=records.device)
    dist.all_gather_into_tensor(slabs, slab)   # the concatenated form: what gloo accepts as well as RCCL
    return GatheredRecords(slabs.view(world, 1 + capacity, width), capacity, world)


def plan_sub_batches(frame_counts, clip_ids, max_frames, max_clips=4096):
    """Cuts one rank's clips into device batches: clips ordered by length (one workgroup walks one clip through all of
    its frames in a single launch of the track kernel and the longest clips are handed out first, so clips of similar
    length leave the shortest tail), a batch closed when it would exceed `max_frames` frames (what fits HBM next to the
    per-frame outputs) or `max_clips` clips.
    -> list of lists of clip ids (every id of `clip_ids` exactly once)."""
    order = sorted((int(i) for i in clip_ids), key=lambda i: (int(frame_counts[i]), i))
    out, cur, cur_frames = [], [], 0
    for i in order:
        n = int(frame_counts[i])
        if cur and (cur_frames + n > max_frames or len(cur) >= max_clips):
            out.append(cur)
            cur, cur_frames = [], 0
        cur.append(i)
        cur_frames += n
    if cur:
        out.append(cur)
    return out


def pack_records(clip_ids, track_ids, scores):
    """Per-track result records for the all-gather: int32 [n, 2 + n_labels] =
    clip id, track id, the float32 class scores bit for bit."""
    import torch

    n = int(scores.shape[0])
    rec = torch.empty((n, 2 + int(scores.shape[1])), dtype=torch.int32, device=scores.device)
    rec[:, 0] = clip_ids.to(torch.int32)
    rec[:, 1] = track_ids.to(torch.int32)
    rec[:, 2:] = scores.contiguous().view(torch.int32)
    return rec


def unpack_records(records):
    """-> (clip ids int32 [n], track ids int32 [n], scores float32 [n, n_labels]) of pack_records / gather_records."""
    import torch

    return records[:, 0], records[:, 1], records[:, 2:].contiguous().view(torch.float32)


def _parse_cpulist(text):
    """'0-15,32-47' -> {0..15, 32..47} (sysfs cpulist format)."""
    cpus = set()
    for part in text.strip().split(","):
        if not part:
            continue
        lo, _, hi = part.partition("-")
        cpus.update(range(int(lo), int(hi or lo) + 1))
    return cpus


def gpu_numa_node(local_rank, sysfs="/sys", visible=None):
    """NUMA node of this rank's GPU, from sysfs alone (no HIP call: it runs before anything touches the GPU): the AMD
    display-class PCI functions under /sys/class/drm/card*/device in PCI-address order are the HIP devices, the
    local_rank-th of them -- after HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES when one is set -- is this rank's.
    -> node id, or None when sysfs does not say (no such card, numa_node = -1: a single-node host)."""
    import glob
    import os

    cards = {}
    for dev in glob.glob(os.path.join(sysfs, "class/drm/card[0-9]*/device")):
        try:
            with open(os.path.join(dev, "vendor")) as fh:
                if fh.read().strip().lower() != "0x1002":
                    continue
            addr = os.path.basename(os.path.realpath(dev))
            with open(os.path.join(dev, "numa_node")) as fh:
                cards[addr] = int(fh.read().strip())
        except (OSError, ValueError):
            continue
    nodes = [cards[a] for a in sorted(cards)]
    if visible is None:
        visible = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if visible:
        try:
            nodes = [nodes[int(v)] for v in visible.split(",") if v.strip() != ""]
        except (ValueError, IndexError):
            return None
    if local_rank < 0 or local_rank >= len(nodes) or nodes[local_rank] < 0:
        return None
    return nodes[local_rank]


def pin_to_gpu_numa(local_rank, sysfs="/sys", apply=True):
    """Keep this rank's host threads (file readers, pinned-memory staging, metadata text: the file-fed path needs ~1.4 CPUs
    per rank, tests/test_host_scaling_cpu.py) on the CPUs of its GPU's NUMA node, so that eight ranks do not stage
    through each other's memory controllers.  os.sched_setaffinity on the calling thread BEFORE any other thread starts
    (threads inherit it); no numactl wrapper, no re-exec.  Nothing happens when sysfs does not name a node, or when the
    node's CPUs and the CPUs this process may use do not intersect.
    -> {"node": id or None, "cpus": CPUs now usable, "pinned": whether the affinity was narrowed}."""
    import os

    try:
        allowed = set(os.sched_getaffinity(0))
    except AttributeError:
        return {"node": None, "cpus": os.cpu_count() or 1, "pinned": False}
    node = gpu_numa_node(local_rank, sysfs)
    out = {"node": node, "cpus": len(allowed), "pinned": False}
    if node is None:
        return out
    try:
        with open(os.path.join(sysfs, "devices/system/node/node%d/cpulist" % node)) as fh:
            want = _parse_cpulist(fh.read()) & allowed
    except (OSError, ValueError):
        return out
    if not want or want == allowed:
        return out
    if apply:
        os.sched_setaffinity(0, want)
    out.update(cpus=len(want), pinned=bool(apply))
    return out
