"""CPU tests (gloo, world_size 2, rendezvous on 127.0.0.1) of the multi-GPU decomposition's host logic and index math.
The kernels are stood in for by the NumPy oracle; what is under test is what the C++ team driver does between kernels:
slab ownership, the [3][G][nxl][nyl][Nzh] pack layout, the all-to-all transpose, the transposed-layout k-space scaling
and the way back, plus the unique-id hand-off helper used to bootstrap RCCL."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, out):
    sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    import torch
    import torch.distributed as dist
    from conftest import make_suspension
    from oracle import pse_port as pp
    from pse_amd.sharded import exchange_unique_id, row_chunks, slab_plan
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        uid = exchange_unique_id(rank, lambda: bytes(range(128)), dist)
        assert uid == bytes(range(128))

        n = 200
        pos, force, box = make_suspension(n, L=16.0, xy=0.25)
        p = pp.select_params(box, 0.5, 1e-3, 0.5, grid=(16, 12, 10))
        Nx, Ny, Nz = p["grid"]; Nzh = Nz // 2 + 1
        plan = slab_plan(p["grid"], world)[rank]
        nxl, nyl, x0, y0 = plan["nxl"], plan["nyl"], plan["x0"], plan["y0"]
        real = pp.spread(pos, force, box, p)                                    # every rank could spread everything ...
        mine = real[:, x0:x0 + nxl]                                             # ... but only owns these planes
        spec2d = np.fft.rfft2(mine, axes=(2, 3))                                # [3][nxl][Ny][Nzh]
        pack = spec2d.reshape(3, nxl, world, nyl, Nzh).transpose(0, 2, 1, 3, 4) # [3][G][nxl][nyl][Nzh]
        recv = np.empty_like(pack)
        for c in range(3):
            a = torch.from_numpy(np.ascontiguousarray(pack[c])); b = torch.empty_like(a)
            dist.all_to_all_single(b, a)
            recv[c] = b.numpy()
        tr = recv.reshape(3, Nx, nyl, Nzh)                                      # [3][Nx][nyl][Nzh], x-major
        tr = np.fft.fft(tr, axis=1)
        kx, ky, kz, k2, w, sinc = pp.kvectors(box, p)
        sl = (slice(None), slice(y0, y0 + nyl), slice(None))
        with np.errstate(divide="ignore", invalid="ignore"):
            kd = (kx[sl] * tr[0] + ky[sl] * tr[1] + kz[sl] * tr[2]) / k2[sl]
        if y0 == 0:
            kd[0, 0, 0] = 0.0
        B = (w * sinc * sinc)[sl]
        tr = np.stack([(tr[0] - kx[sl] * kd) * B, (tr[1] - ky[sl] * kd) * B, (tr[2] - kz[sl] * kd) * B])
        tr = np.fft.ifft(tr, axis=1) * Nx                                       # unnormalised inverse
        back = np.empty_like(pack)
        send = tr.reshape(3, world, nxl, nyl, Nzh)
        for c in range(3):
            a = torch.from_numpy(np.ascontiguousarray(send[c])); b = torch.empty_like(a)
            dist.all_to_all_single(b, a)
            back[c] = b.numpy()
        spec_back = back.transpose(0, 2, 1, 3, 4).reshape(3, nxl, Ny, Nzh)
        ug = np.fft.irfft2(spec_back, s=(Ny, Nz), axes=(2, 3)) * (Ny * Nz)
        full = np.fft.irfftn(pp.wave_scale(np.fft.rfftn(real, axes=(1, 2, 3)), box, p), s=p["grid"], axes=(1, 2, 3), norm="forward")
        err = np.abs(ug - full[:, x0:x0 + nxl]).max() / np.abs(full).max()
        assert err < 1e-12, err

        # gather ownership: every particle belongs to exactly one slab (the one holding its support origin)
        idx, dlt = pp._support(pos, box, p)
        start = (idx[0][:, 0] - 0) % Nx
        owned = ((start - x0) % Nx) < nxl
        cnt = torch.tensor([int(owned.sum())]); dist.all_reduce(cnt)
        assert int(cnt) == n
        # row chunks tile [0, n)
        ch = row_chunks(n, world)
        assert ch[0][0] == 0 and ch[-1][1] == n and all(ch[i][1] == ch[i + 1][0] for i in range(world - 1))
        out.put((rank, "ok"))
    except Exception as e:   # noqa: BLE001
        out.put((rank, repr(e)))
    finally:
        dist.destroy_process_group()


def test_slab_transpose_pipeline_world2():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    out = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    res = [out.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    assert sorted(res) == [(0, "ok"), (1, "ok")], res


def test_slab_plan_and_row_chunks():
    from pse_amd.sharded import row_chunks, slab_plan
    pl = slab_plan((256, 256, 256), 8)
    assert [p["x0"] for p in pl] == list(range(0, 256, 32)) and all(p["nxl"] == 32 and p["nyl"] == 32 for p in pl)
    with pytest.raises(ValueError):
        slab_plan((250, 256, 256), 8)
    assert row_chunks(10, 4) == [(0, 3), (3, 6), (6, 9), (9, 10)]
    assert row_chunks(1_000_000, 8)[-1] == (875_000, 1_000_000)
