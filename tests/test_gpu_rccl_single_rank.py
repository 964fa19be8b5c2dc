"""The N>1 step of bench.py on one GPU: bin shard (world 1), RCCL all-reduce of the packed
level fluxes through a torch tensor aliasing the library's buffer, ordered on the library's own
HIP stream, then radtran_finish_reduced.  One rank is all a one-GPU box allows; it still drives
the alias, the stream hand-off and RCCL itself."""
import os
import socket

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def test_allreduce_on_library_stream(hip_lib, small_tables):
    import torch
    import torch.distributed as dist
    from clima_amd import synthetic as S
    from clima_amd.radtran import Radtran

    nz = 40
    col = S.modern_earth_column(nz)
    r = Radtran(small_tables, nz, 2, 0.25)
    r.radiate(*col.args())
    want = np.array(r.f_total)
    want_flux = np.concatenate([r.wrk_ir.fup_n, r.wrk_ir.fdn_n, r.wrk_sol.fup_n, r.wrk_sol.fdn_n])

    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(_free_port())
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        r.set_bin_shard(0, 1)
        r.upload_column(*col.args())
        flux = r.flux_tensor()
        assert flux.is_cuda and flux.dtype == torch.float64 and flux.numel() == 4 * (nz + 1)
        assert flux.data_ptr() == r.flux_device_ptr()[0]            # an alias, not a copy
        stream = torch.cuda.ExternalStream(r.stream())
        for _ in range(3):
            r.radiate_resident()
            with torch.cuda.stream(stream):
                dist.all_reduce(flux)
            r.finish_reduced()
        r.synchronize()
        torch.cuda.synchronize()
        np.testing.assert_array_equal(flux.cpu().numpy(), want_flux)
        np.testing.assert_array_equal(np.array(r.f_total), want)
        # host-synchronised form
        r.radiate_resident()
        r.synchronize()
        dist.all_reduce(flux)
        torch.cuda.current_stream().synchronize()
        r.finish_reduced()
        np.testing.assert_array_equal(np.array(r.f_total), want)
    finally:
        dist.destroy_process_group()
