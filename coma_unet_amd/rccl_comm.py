"""RCCL communicator behind the C ABI (include/coma_unet.h, ``coma_comm_*`` / ``coma_allreduce_sum_f32`` ...).

One communicator per process (= per GPU).  The 128-byte RCCL id is created by rank 0 through the library and handed to
the other ranks over the torch.distributed group the launcher already set up (any backend: it is a host-side byte string);
after that every collective of the data-parallel step goes through the C ABI on a HIP stream of the caller's choice --
a side stream, so that a gradient bucket's exchange overlaps the rest of backward, and so that the whole step including
its collectives can be captured in ONE hipGraph (train.GraphedTrainStep, ``COMA_DP=capi``).
"""
from __future__ import annotations

import ctypes

import torch
import torch.distributed as dist

from ._lib import lib, check


class RcclComm:
    def __init__(self, group=None, device=None):
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.device = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
        ident = ctypes.create_string_buffer(128)
        if self.rank == 0:
            check(lib.coma_comm_unique_id(ident), "coma_comm_unique_id")
        if self.world > 1:
            box = [bytes(ident.raw)]
            dist.broadcast_object_list(box, src=0, group=group)
            ident = ctypes.create_string_buffer(box[0], 128)
        handle = ctypes.c_void_p()
        with torch.cuda.device(self.device):
            check(lib.coma_comm_init(ident, self.rank, self.world, ctypes.byref(handle)), "coma_comm_init")
        self._h = handle
        self.side = torch.cuda.Stream(device=self.device)     # the exchange stream

    def _chk(self, t):
        assert t.is_cuda and t.dtype == torch.float32 and t.is_contiguous(), "fp32 contiguous device buffers only"

    def all_reduce_(self, t, stream=None):
        self._chk(t)
        s = (stream or torch.cuda.current_stream()).cuda_stream
        check(lib.coma_allreduce_sum_f32(self._h, t.data_ptr(), t.numel(), s), "coma_allreduce_sum_f32")

    def reduce_scatter(self, send, recv, stream=None):
        self._chk(send), self._chk(recv)
        assert send.numel() == recv.numel() * self.world
        s = (stream or torch.cuda.current_stream()).cuda_stream
        check(lib.coma_reduce_scatter_sum_f32(self._h, send.data_ptr(), recv.data_ptr(), recv.numel(), s), "coma_reduce_scatter_sum_f32")

    def all_gather(self, send, recv, stream=None):
        self._chk(send), self._chk(recv)
        assert recv.numel() == send.numel() * self.world
        s = (stream or torch.cuda.current_stream()).cuda_stream
        check(lib.coma_allgather_f32(self._h, send.data_ptr(), recv.data_ptr(), send.numel(), s), "coma_allgather_f32")

    def broadcast_(self, t, root=0, stream=None):
        self._chk(t)
        s = (stream or torch.cuda.current_stream()).cuda_stream
        check(lib.coma_broadcast_f32(self._h, t.data_ptr(), t.numel(), root, s), "coma_broadcast_f32")

    def close(self):
        if self._h:
            torch.cuda.synchronize(self.device)
            check(lib.coma_comm_destroy(self._h), "coma_comm_destroy")
            self._h = None
