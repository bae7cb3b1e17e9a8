"""Sharding of the fit across the GPUs of one node.

One process per GPU (`torch.distributed`, backend "nccl" = RCCL over xGMI).  The
fit statistics are sums over frames, so any partition of the recordings works:
every rank accumulates its own files, then ONE all-reduce(sum) of the packed
float64 buffer gives every rank the global statistics.  The buffer holds the
compact lag statistics (additive) followed by one slot per file for the
file-boundary samples; a rank fills only its own slots, so the sum over ranks
is the concatenation -- no second collective is needed.

The reference has no communication layer at all (process-level fan-out only,
doc/DecodingCodelab.md:354-381); this module is new.
"""
import os

import numpy as np


def _dist():
  import torch.distributed as dist
  return dist


class ShardPlan(object):
  """Which files go to which rank, and where their boundary slots live.

  Files are dealt in contiguous runs balanced by frame count (greedy), so a
  rank's files stay one concatenated array.
  """

  def __init__(self, file_lengths, world_size):
    self.file_lengths = [int(n) for n in file_lengths]
    self.world_size = int(world_size)
    total = float(sum(self.file_lengths)) or 1.0
    owner = []
    before = 0.0
    for n in self.file_lengths:       # a file goes to the rank its midpoint falls in
      mid = (before + n / 2.0) / total
      owner.append(min(self.world_size - 1, int(mid * self.world_size)))
      before += n
    self.bounds = [0] * (self.world_size + 1)
    for r in range(self.world_size):
      self.bounds[r + 1] = self.bounds[r] + sum(1 for o in owner if o == r)

  @property
  def total_files(self):
    return len(self.file_lengths)

  def files_of(self, rank):
    return list(range(self.bounds[rank], self.bounds[rank + 1]))

  def slot_of(self, rank):
    """First boundary slot of `rank` in the packed buffer."""
    return self.bounds[rank]

  def frames_of(self, rank):
    return sum(self.file_lengths[f] for f in self.files_of(rank))


class TimeShardPlan(object):
  """Strong scaling of ONE fit: the concatenated recordings are cut into `world_size`
  contiguous time ranges of (nearly) equal length, so a long recording is shared by several
  ranks (third unit).  Rank r sums the rows of its range; of every recording
  it touches it holds a PIECE = the range plus a read-only halo of `halo` rows on either side
  (clipped at the recording's true ends), so lagged products across the cut see real data and
  the zero extension happens only at true ends.  Boundary slots of the packed all-reduce
  buffer are per recording: the ranks sharing one map it to the same slot, only the piece
  that holds an end contributes that end's boundary window.

  halo must be >= pre + post of every lagged input (device.LagStats.hw, which also covers the
  boundary windows, is the safe choice)."""

  def __init__(self, file_lengths, world_size, halo, batch_size=None):
    self.file_lengths = [int(n) for n in file_lengths]
    self.world_size = int(world_size)
    self.halo = int(halo)
    # rows that enter the fit: batch(drop_remainder=True) drops the tail of the LAST file
    used = list(self.file_lengths)
    if batch_size:
      total = sum(used)
      used[-1] -= total % int(batch_size)
      if used[-1] < 0:
        raise ValueError('the last recording is shorter than the dropped remainder')
    self.rows_used = used
    total = sum(used)
    self.cuts = [(total * r) // self.world_size for r in range(self.world_size + 1)]
    self.starts = np.concatenate(([0], np.cumsum(used))).astype(np.int64)   # in used rows

  @property
  def total_files(self):
    return len(self.file_lengths)

  @property
  def total_frames(self):
    return int(sum(self.rows_used))

  def pieces_of(self, rank):
    """[(file, piece_first, piece_last, range_begin, range_end, edge_flags, rows_used)]: the
    piece is rows [piece_first, piece_last) of the recording; the range and rows_used are in
    PIECE coordinates."""
    lo, hi = self.cuts[rank], self.cuts[rank + 1]
    out = []
    for f, n in enumerate(self.file_lengths):
      a = max(lo, int(self.starts[f])) - int(self.starts[f])
      b = min(hi, int(self.starts[f + 1])) - int(self.starts[f])
      if b <= a:
        continue
      first = max(a - self.halo, 0)
      last = min(b + self.halo, n)
      # an end's boundary window comes from the ONE piece whose range touches that end
      flags = (1 if a == 0 else 0) | (2 if b == self.rows_used[f] else 0)
      # the piece's N': the recording's rows_used seen from the piece (only the tail piece is
      # ever cut by it; ranges never reach past it)
      used = min(self.rows_used[f], last) - first
      out.append((f, first, last, a - first, b - first, flags, used))
    return out

  def slot_of(self, rank):
    pieces = self.pieces_of(rank)
    return pieces[0][0] if pieces else 0

  def frames_of(self, rank):
    return self.cuts[rank + 1] - self.cuts[rank]


def accumulate_time_shard(stats, plan, rank, arrays, handle=None, parts=3):
  """Adds rank `rank`'s pieces to `stats`.  arrays(file, first, last) -> (x, x2, y) device
  tensors holding rows [first, last) of that recording (x2 / y may be None)."""
  # the half width of the statistics' boundary windows (td_stats: hw), which covers the
  # context of both lagged inputs
  need = stats.pre1 + stats.post1 + stats.pre2 + stats.post2 + 1
  if plan.world_size > 1 and plan.halo < need:
    raise ValueError('TimeShardPlan halo of %d rows does not cover the context of the statistics '
                     '(%d rows needed): lagged products across a cut would see zeros instead of '
                     'the neighbouring rows' % (plan.halo, need))
  pieces = plan.pieces_of(rank)
  if not pieces:
    return stats
  import torch
  xs, x2s, ys, offs = [], [], [], [0]
  for f, first, last, _, _, _, _ in pieces:
    x, x2, y = arrays(f, first, last)
    xs.append(x); x2s.append(x2); ys.append(y)
    offs.append(offs[-1] + last - first)
  cat = lambda parts_: None if parts_[0] is None else (
      parts_[0] if len(parts_) == 1 else torch.cat(parts_).contiguous())
  stats.accumulate(cat(xs), cat(x2s), cat(ys), offs, rows_used=[p[6] for p in pieces],
                   ranges=[(p[3], p[4]) for p in pieces], edges=[p[5] for p in pieces],
                   handle=handle, parts=parts)
  return stats


class RcclComm(object):
  """An RCCL communicator owned by the C-ABI (td_rccl_comm_create), one rank per process:
  what td_stats_allreduce / td_allreduce_f64 run on.  Rank 0 makes the 128-byte id
  (ncclGetUniqueId), `exchange(id_bytes) -> id_bytes` hands it to the other ranks by any
  means (here: a torch.distributed broadcast, which works over gloo and over RCCL alike)."""

  def __init__(self, handle, rank, world_size, exchange):
    import ctypes
    self.h, self.rank, self.world_size = handle, int(rank), int(world_size)
    ident = (ctypes.c_char * 128)()
    if self.rank == 0:
      handle.check(handle.lib.td_rccl_unique_id(handle.ptr, ident))
    ident = (ctypes.c_char * 128).from_buffer_copy(exchange(bytes(ident.raw)))
    ptr = ctypes.c_void_p()
    handle.check(handle.lib.td_rccl_comm_create(handle.ptr, self.world_size, self.rank, ident,
                                                ctypes.byref(ptr)))
    self.ptr = ptr
    n = ctypes.c_int(0)
    handle.check(handle.lib.td_rccl_comm_count(handle.ptr, self.ptr, ctypes.byref(n)))
    if n.value != self.world_size:
      raise RuntimeError('RCCL reports %d ranks, expected %d' % (n.value, self.world_size))

  def close(self):
    if getattr(self, 'ptr', None):
      self.h.lib.td_rccl_comm_destroy(None, self.ptr)    # (the handle may be gone by now)
      self.ptr = None

  def __del__(self):
    try:
      self.close()
    except Exception:  # interpreter shutdown
      pass


_comms = {}                 # (id(group), ranks of the group, device) -> (communicator or None, group)
LAST_COLLECTIVE = {}        # what the last all-reduce of this process went through (bench.py reports it)


def _comm_key(dist, handle, group):
  """The cache key of a process group: its identity AND its membership.  id() alone can be reused by
  a new group once the old object is collected (the new group would inherit a communicator with the
  wrong ranks and hang); the entry also keeps the group object alive, so the id cannot come back
  while the communicator is cached."""
  ranks = tuple(dist.get_process_group_ranks(group)) if group is not None else \
      tuple(range(dist.get_world_size()))
  return (id(group) if group is not None else 0, ranks, handle.device_id)


def init_native_comm(handle, group=None):
  """Creates the C-ABI communicator of `group` NOW.  Creation is itself collective (two agreement
  all-reduces, a broadcast of the id, ncclCommInitRank): every rank of the group must call this at
  the same point of the program -- right after init_process_group / new_group is the place.
  (allreduce_stats / allreduce_packed create it lazily at their first use, which is only safe when
  every rank reaches that first use together.)  Returns the communicator, or None when the exchange
  goes through torch.distributed (no process group, gloo, TD_ALLREDUCE_TORCH, librccl missing)."""
  return native_comm(handle, group)


def native_comm(handle, group=None):
  """The C-ABI's own communicator over the ranks of `group` (created once per process group, the
  id broadcast through torch.distributed), or None when the exchange has to go through
  torch.distributed itself: no process group, a CPU (gloo) group -- the CPU tests' NumPy device
  stand-in -- or TD_ALLREDUCE_TORCH set (A/B switch)."""
  dist = _dist()
  if not (dist.is_available() and dist.is_initialized()) or os.environ.get('TD_ALLREDUCE_TORCH'):
    return None
  if not hasattr(handle, 'lib') or dist.get_backend(group) != 'nccl':
    return None
  world = dist.get_world_size(group)
  if world == 1 and not os.environ.get('TD_ALLREDUCE_ALWAYS'):
    return None
  key = _comm_key(dist, handle, group)
  if key not in _comms:
    import sys
    import torch

    def agree(ok):
      """True only if every rank of the group says so (the ranks must take the same route)."""
      t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=handle.device)
      dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
      return bool(t.item())

    def exchange(ident):
      t = torch.tensor(list(ident), dtype=torch.uint8, device=handle.device)
      dist.broadcast(t, src=dist.get_global_rank(group, 0) if group is not None else 0, group=group)
      return bytes(t.cpu().tolist())

    # (1) can every rank bind librccl at all?  (dlopen + dlsym: no RCCL call, no bootstrap thread)
    usable = handle.lib.td_rccl_available(handle.ptr) == _lib_ok()
    comm = None
    if agree(usable):
      # (2) the communicator itself (ncclCommInitRank: a collective of its own), then agree again
      err = None
      try:
        comm = RcclComm(handle, dist.get_rank(group), world, exchange)
      except Exception as e:            # pylint: disable=broad-except
        err = e
      if not agree(comm is not None):
        if comm is not None:
          comm.close()
        comm = None
        if err is not None:
          sys.stderr.write('telluride_decoding_amd: C-ABI communicator not created (%s); the '
                           'statistics all-reduce goes through torch.distributed\n' % err)
    else:
      sys.stderr.write('telluride_decoding_amd: librccl could not be bound on every rank; the '
                       'statistics all-reduce goes through torch.distributed\n')
    _comms[key] = (comm, group)
  return _comms[key][0]


def _lib_ok():
  from telluride_decoding_amd import _lib
  return _lib.TD_OK


def close_native_comms():
  """Destroys the C-ABI communicators (before torch.distributed.destroy_process_group)."""
  for c, _ in _comms.values():
    if c is not None:
      c.close()
  _comms.clear()


def allreduce_packed(buf, group=None, handle=None):
  """Sum a packed statistics buffer over ranks, in place: td_allreduce_f64 over the C-ABI's
  RCCL communicator for device buffers (on `handle`'s stream), torch.distributed (gloo) for
  the CPU tensors of the tests."""
  dist = _dist()
  if dist.is_available() and dist.is_initialized() and (
      dist.get_world_size(group) > 1 or os.environ.get('TD_ALLREDUCE_ALWAYS')):
    comm = native_comm(handle, group) if (handle is not None and buf.is_cuda) else None
    if comm is not None:
      import ctypes
      if str(buf.dtype) != 'torch.float64' or not buf.is_contiguous():
        raise TypeError('the packed buffer must be a contiguous float64 tensor')
      handle.check(handle.lib.td_allreduce_f64(handle.ptr, ctypes.c_void_p(buf.data_ptr()),
                                               buf.numel(), comm.ptr))
      LAST_COLLECTIVE.update(route='td_allreduce_f64', ranks=comm.world_size)
    else:
      dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
      LAST_COLLECTIVE.update(route='torch.distributed', ranks=dist.get_world_size(group))
  return buf


def allreduce_stats(stats, plan, rank, group=None, total_frames=None, handle=None):
  """Every rank ends with the statistics of all files (one all-reduce).  total_frames: the
  frames of all ranks if known on the host (no dropped remainders / offsets: the sum of the
  file lengths) -- spares the unpack a stream synchronisation, which matters when fits are
  pipelined.  handle: queue pack / collective / unpack on that handle's stream instead of the
  statistics' own.

  On the device this is ONE C-ABI call, td_stats_allreduce (pack -> ncclAllReduce -> unpack on
  the handle's stream); under gloo (CPU tests, NumPy stand-in of the device
  layer) the same three steps go through torch.distributed."""
  h = handle or getattr(stats, 'h', None)
  comm = native_comm(h, group) if h is not None else None
  if comm is not None:
    h.check(h.lib.td_stats_allreduce(
        h.ptr, stats.ptr, comm.ptr, int(plan.total_files), int(plan.slot_of(rank)),
        -1 if total_frames is None else int(total_frames)))
    LAST_COLLECTIVE.update(route='td_stats_allreduce', ranks=comm.world_size)
    return stats
  buf = stats.pack(plan.total_files, plan.slot_of(rank), handle=handle)
  allreduce_packed(buf, group)
  stats.unpack(buf, plan.total_files, total_frames, handle=handle)
  return stats


def split_round_robin(items, rank, world_size):
  """Embarrassingly parallel work (folds, lambdas): item i goes to rank i % world."""
  return [it for i, it in enumerate(items) if i % world_size == rank]


def gather_rows(local_rows, n_total, index, group=None, local_only=False):
  """Each rank computed rows `index` of an [n_total, ...] result; returns the full
  array on every rank (sum of disjoint contributions).  local_only: this caller holds every row
  (a one-rank computation inside a multi-rank job): no collective."""
  import torch
  local_rows = np.asarray(local_rows, np.float64)
  full = np.zeros((n_total,) + local_rows.shape[1:], np.float64)
  for j, i in enumerate(index):
    full[i] = local_rows[j]
  dist = _dist()
  if not local_only and dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
    t = torch.from_numpy(full)
    if dist.get_backend(group) == 'nccl':
      t = t.cuda()
    dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    full = t.cpu().numpy()
  return full
