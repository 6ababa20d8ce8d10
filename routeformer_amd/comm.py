"""Host side of ``rf_comm_*`` (csrc/comm.hip, include/rf_hip.h): the data-parallel gradient exchange as plain RCCL calls on
an explicit communication stream gated by HIP events -- what Lightning's ``DDPStrategy(process_group_backend="nccl")``
does for the reference (experiments/full_comparison.py:794,832), without a process-group object between the engine and
RCCL.  ``GradReducer`` uses it when ``RF_DP_COMM=rf`` (default: torch.distributed's ProcessGroupNCCL, which is RCCL too).

The 128-byte RCCL unique id is the only thing that has to travel between ranks before the communicator exists; it goes
over whatever host channel is at hand: an initialised ``torch.distributed`` group (any backend -- gloo is enough), or the
caller passes the bytes."""
from __future__ import annotations

import ctypes
from typing import Optional

import torch

from . import _hip
from ._hip import check


class RfComm:
    def __init__(self, rank: int = 0, world: int = 1, unique_id: Optional[bytes] = None, group=None):
        """Collective over the ``world`` ranks (each with its GPU already selected: ``torch.cuda.set_device``)."""
        lib = _hip.lib()
        if not lib.rf_comm_available():
            raise _hip.HipLibraryError("rf_comm: RCCL could not be resolved in this process")
        if unique_id is None:
            buf = (ctypes.c_char * 128)()
            if rank == 0:
                check(lib.rf_comm_unique_id(buf), "rf_comm_unique_id")
            if world > 1:
                import torch.distributed as dist
                assert dist.is_initialized(), "rf_comm: pass unique_id or initialise torch.distributed (gloo is enough)"
                box = [bytes(buf) if rank == 0 else None]
                dist.broadcast_object_list(box, src=0, group=group)
                unique_id = box[0]
            else:
                unique_id = bytes(buf)
        assert len(unique_id) == 128
        self.rank, self.world = rank, world
        self._h = ctypes.c_void_p()
        self._id = ctypes.create_string_buffer(unique_id, 128)
        check(lib.rf_comm_init(ctypes.byref(self._h), self._id, rank, world), "rf_comm_init")

    @staticmethod
    def _stream(stream=None):
        return (stream or torch.cuda.current_stream()).cuda_stream

    def allreduce_bucket(self, t: torch.Tensor, average: bool = False, stream=None):
        """In-place all-reduce of a contiguous fp32 / bf16 device tensor, ordered after ``stream`` (default: current)."""
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        check(_hip.lib().rf_comm_allreduce_bucket(self._h, t.data_ptr(), t.numel(), 0 if t.dtype == torch.float32 else 1,
                                                   1 if average else 0, self._stream(stream)), "rf_comm_allreduce_bucket")

    def broadcast(self, t: torch.Tensor, root: int = 0, stream=None):
        assert t.is_cuda and t.is_contiguous() and t.dtype in (torch.float32, torch.bfloat16)
        check(_hip.lib().rf_comm_broadcast(self._h, t.data_ptr(), t.numel(), 0 if t.dtype == torch.float32 else 1, root,
                                            self._stream(stream)), "rf_comm_broadcast")

    def wait(self, stream=None):
        """``stream`` (default: current) waits on the device for every collective launched so far."""
        check(_hip.lib().rf_comm_wait(self._h, self._stream(stream)), "rf_comm_wait")

    def close(self):
        if self._h:
            _hip.lib().rf_comm_destroy(self._h)
            self._h = ctypes.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass
