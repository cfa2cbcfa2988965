"""The C-ABI collectives (isx_comm_* / isx_allreduce_grads / isx_gather_tiles / isx_allreduce_max_labels, comm.hip) on the one
GPU of the test box: a world of one rank exercises the whole path that does not depend on a peer -- lazy loading of RCCL,
communicator creation on the context's device, the calls on the context's stream, the 1/world scaling, teardown -- and the
error behaviour.  More ranks need more GPUs (RCCL refuses two ranks on one device); the multi-rank exchanges themselves are
covered through torch.distributed in tests/test_distributed_gloo.py and the two-rank GPU workers."""
import ctypes as C

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_single_rank_communicator_roundtrip():
    assert torch.cuda.is_available(), "these tests need the MI355X"
    from iterseg_amd import _lib
    ctx = _lib.Context(0)
    lib, h = ctx.lib, ctx.handle
    ctx.use_torch_stream()
    g = torch.arange(1000, dtype=torch.float32, device="cuda") * 0.5 - 3.0
    # before a communicator exists: a state error, nothing touched
    assert lib.isx_allreduce_grads(h, C.c_void_p(g.data_ptr()), g.numel()) == _lib.ISX_ERR_STATE
    ident = (C.c_uint8 * 128)()
    _lib.check(lib.isx_comm_unique_id(ident))
    assert any(ident)                                                       # an ncclUniqueId, not zeros
    _lib.check(lib.isx_comm_init(h, ident, 0, 1))
    assert lib.isx_comm_init(h, ident, 0, 1) != 0                            # one communicator per context
    ref = g.clone()
    _lib.check(lib.isx_allreduce_grads(h, C.c_void_p(g.data_ptr()), g.numel()))
    torch.cuda.synchronize()
    assert torch.equal(g, ref)                                               # sum over one rank, divided by one
    send = torch.randint(0, 255, (4096,), dtype=torch.uint8, device="cuda")
    recv = torch.zeros_like(send)
    _lib.check(lib.isx_gather_tiles(h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel()))
    lab = torch.randint(0, 1 << 20, (5000,), dtype=torch.int32, device="cuda")
    lref = lab.clone()
    _lib.check(lib.isx_allreduce_max_labels(h, C.c_void_p(lab.data_ptr()), lab.numel()))
    torch.cuda.synchronize()
    assert torch.equal(recv, send) and torch.equal(lab, lref)
    _lib.check(lib.isx_comm_destroy(h))
    assert lib.isx_gather_tiles(h, C.c_void_p(send.data_ptr()), C.c_void_p(recv.data_ptr()), send.numel()) == _lib.ISX_ERR_STATE
    # bad rank / world
    assert lib.isx_comm_init(h, ident, 2, 2) != 0
    ctx.close()


def test_context_teardown_releases_the_communicator():
    from iterseg_amd import _lib
    ctx = _lib.Context(0)
    ident = (C.c_uint8 * 128)()
    _lib.check(ctx.lib.isx_comm_unique_id(ident))
    _lib.check(ctx.lib.isx_comm_init(ctx.handle, ident, 0, 1))
    ctx.close()                                                              # isx_ctx_destroy destroys the communicator too
    ctx2 = _lib.Context(0)
    _lib.check(ctx2.lib.isx_comm_unique_id(ident))
    _lib.check(ctx2.lib.isx_comm_init(ctx2.handle, ident, 0, 1))
    ctx2.close()
