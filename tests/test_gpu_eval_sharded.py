"""GPU, 2 processes sharing the card, gloo backend (RCCL refuses two ranks on one device): validateModels.validate_sharded -- every rank
extracts the features of ITS gallery slice, computes its [Nq, Ng / 2] distance block, and the ranks merge per-query hit counts
(ops_eval.rank_eval_sharded: all-gather of match keys, all-reduce of integer bins) -- against the single-process validate() on the whole
gallery: CMC and mAP bit-identical, on every rank (validateModels.py:35-76)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from conftest import ROOT

pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _setup():
    from daliid_amd import Encoders, synthetic, validateModels
    data = synthetic.SyntheticImages(n_ids=24, per_id=9, n_cams=3, seed=5, noise=1.2).install()
    online = Encoders._DataParallelShim(Encoders.ResNet50ReID(layers=(1, 1, 1, 1), width=32, seed=9))
    _, gallery, query = data.split(2)
    v = validateModels.validationManager.getValidator("Market")
    v.setParameters(64, 32, False, 0)
    return v, online.eval(), query, gallery


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
    import contextlib, io
    import torch.distributed as dist
    from daliid_amd import parallel
    torch.cuda.set_device(0)
    parallel.init_from_env("gloo")
    v, online, query, gallery = _setup()
    with contextlib.redirect_stdout(io.StringIO()):
        cmc, mAP, block = v.validate_sharded(query, gallery, online, dist.group.WORLD)
    torch.save({"cmc": cmc, "mAP": mAP, "block": tuple(block.shape)}, os.path.join(out_dir, "rank%d.pt" % rank))
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_sharded_validate_equals_single_rank(tmp_path):
    if not torch.cuda.is_available():
        pytest.skip("needs a GPU")
    world = 2
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    outs = [torch.load(os.path.join(str(tmp_path), "rank%d.pt" % r), weights_only=False) for r in range(world)]
    v, online, query, gallery = _setup()
    try:
        cmc, mAP, distmat = v.validate(query, gallery, online)
    finally:
        from daliid_amd import synthetic
        synthetic.SyntheticImages.uninstall()
    assert 0.0 < mAP <= 1.0
    from daliid_amd import ops_eval
    b = ops_eval.shard_bounds(len(gallery), world)
    for r in range(world):
        assert outs[r]["block"] == (len(query), b[r + 1] - b[r])
        assert np.array_equal(outs[r]["cmc"], cmc) and outs[r]["mAP"] == mAP, (r, outs[r]["mAP"], mAP)
