"""Data parallelism for the QuadtreeCNN train step: one process per GPU,
replicated weights, each rank runs the plan on its shard of the batch, one
gradient all-reduce (RCCL over xGMI; `nccl` backend of torch.distributed) per
step.  The reference has no distributed code at all; message
sizes and overlap plan are in SURVEY.md 8(e).

The plan's backward runs in four phases and hands over its flat f32 gradient buffer
in four buckets, in the order they become final (engine.py::PlanEngine.backward):
  head      classifier + numerical MLP + quadrant conv  (59 MB, 57 % of the bytes,
            ready before any backbone kernel has run)
  layer4    33.6 MB, ready after the first two residual blocks of backward
  layer3+2  10.5 MB, on the wire while layer1 and the stem run
  layer1+stem  0.6 MB: the only bucket whose reduction is exposed behind the last kernel
Each bucket is averaged across ranks on a dedicated communication stream while
the compute stream continues with the backbone backward; the compute stream
waits for the communication stream once, at the end of backward.  BatchNorm
statistics stay per replica (the reference has no SyncBN).
"""
import torch
import torch.distributed as dist


def shard_range(global_batch, rank, world_size):
    """[begin, end) of the samples of `rank`: contiguous, sizes differ by at most one."""
    base, rem = divmod(int(global_batch), int(world_size))
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


class GradBucketReducer:
    """callable(bucket, phase): average `bucket` over the process group.
    phase 0 / bucket None = join (make the results visible to the compute stream)."""

    def __init__(self, process_group=None, average=True):
        self.group = process_group
        self.average = average
        self.world = dist.get_world_size(process_group)
        self.comm_stream = None
        self.bytes_reduced = 0
        self.force = False  # exercise the collective even with one rank (rehearsal)
        self.bucket_log = []        # (phase, bytes) of the first step's buckets, in hand-over order
        self.buckets_per_step = 0
        self._first_step_open = True
        self._avg_ok = dist.get_backend(process_group) == "nccl"
        # exposed tail of the reduction: with `measure_tail` set, every join brackets its wait for the communication stream
        # with two events on the compute stream -- the time the last compute kernel has been finished while the all-reduce
        # of the last bucket(s) was still running.  (bench.py sets it for its timed steps and reports the mean.)
        self.measure_tail = False
        self._tail_events = []

    def __call__(self, bucket, phase, side_fence=None):
        if bucket is None:
            if self._first_step_open and self.bucket_log:
                self._first_step_open = False
                self.buckets_per_step = len(self.bucket_log)
            if self.comm_stream is not None:
                cur = torch.cuda.current_stream()
                if self.measure_tail:
                    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    a.record(cur)
                    cur.wait_stream(self.comm_stream)
                    b.record(cur)
                    self._tail_events.append((a, b))
                else:
                    cur.wait_stream(self.comm_stream)
            return
        self.bytes_reduced += bucket.numel() * bucket.element_size()
        if self._first_step_open:
            self.bucket_log.append([int(phase), int(bucket.numel() * bucket.element_size())])
        if self.world == 1 and not self.force:
            return
        if bucket.is_cuda:
            if self.comm_stream is None:
                self.comm_stream = torch.cuda.Stream(device=bucket.device)
            self.comm_stream.wait_stream(torch.cuda.current_stream())
            if side_fence is not None:   # the bucket's weight gradients come from the plan's side stream
                side_fence(self.comm_stream)
            bucket.record_stream(self.comm_stream)
            with torch.cuda.stream(self.comm_stream):
                if self.average and self._avg_ok:
                    dist.all_reduce(bucket, op=dist.ReduceOp.AVG, group=self.group)
                else:
                    dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
                    if self.average:
                        bucket.div_(self.world)
        else:  # gloo on CPU tensors (tests)
            dist.all_reduce(bucket, op=dist.ReduceOp.SUM, group=self.group)
            if self.average:
                bucket.div_(self.world)

    def exposed_tail_ms(self):
        """Mean wait of the compute stream at the join over the measured steps (synchronise first), None if none."""
        if not self._tail_events:
            return None
        ms = [a.elapsed_time(b) for a, b in self._tail_events]
        self._tail_events = []
        return sum(ms) / len(ms)


def broadcast_state(model, src=0, process_group=None):
    """Replicate rank `src`'s parameters and buffers (aliased tensors once)."""
    seen = set()
    for t in list(model.parameters()) + list(model.buffers()):
        if t.data_ptr() in seen:
            continue
        seen.add(t.data_ptr())
        dist.broadcast(t.data, src=src, group=process_group)


def attach_data_parallel(model, process_group=None, broadcast=True):
    """Make `model` (a QuadtreeCNN / StandardResNetCNN of this package) average its
    gradients over the process group inside backward().  The training loop itself
    (reference: Quadtree_from scratch/Quadtree_train.py:60-66) stays unchanged."""
    if broadcast:
        broadcast_state(model, 0, process_group)
    model._grad_sync = GradBucketReducer(process_group)
    import os
    model._grad_sync.force = os.environ.get("QTCNN_FORCE_DIST") == "1"  # "2": phases only, no collective
    if getattr(model, "_engine", None) is not None:
        model._engine.grad_sync = model._grad_sync
    return model
