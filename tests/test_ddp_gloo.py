"""World-size-2 data-parallel step on the CPU (gloo): the trainer's DDP wrapping and train_step harness, driven with
an oracle-backed stand-in for the model (the HIP product path has no CPU mode; tests may use the oracle).
Checks: (1) every rank ends the step with identical parameters, (2) the all-reduced gradient equals the mean of the
ranks' local gradients, (3) the loss dict contract."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CFG = dict(volume_size=(16, 16, 16), xray_img_size=64, voxel_dim=32, vit_depth=1, num_heads=1, xray_feature_dim=32)


def _build():
    from direct_regression.model_direct import DirectCTRegression
    from oracle import hvc_oracle as O

    class OracleBackedDirect(torch.nn.Module):
        """Parameters / buffers of the product module, arithmetic of the CPU oracle."""

        def __init__(self):
            super().__init__()
            torch.manual_seed(0)
            self.inner = DirectCTRegression(**CFG)
            g = torch.Generator().manual_seed(5)
            with torch.no_grad():
                for blk in self.inner.vit_backbone.blocks:
                    blk.adaln.linear.weight.copy_(torch.randn(blk.adaln.linear.weight.shape, generator=g) * 0.02)

        def forward(self, xrays):
            P = dict(self.inner.named_parameters())
            P.update(dict(self.inner.named_buffers()))
            return O.direct_ct_regression(xrays, P, CFG["volume_size"], CFG["voxel_dim"], CFG["vit_depth"], CFG["num_heads"],
                                          training=False)

    class OracleLoss(torch.nn.Module):
        def forward(self, pred, target):
            return O.direct_regression_loss(pred.float(), target)

    return OracleBackedDirect(), OracleLoss()


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "hybrid-vit-cascade_amd"))
    from direct_regression import train_direct_4gpu as T
    from hvc import synthetic
    torch.set_num_threads(2)
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    T.setup_ddp(rank, world, backend="gloo", port=str(port))
    model, crit = _build()
    xr, ct = synthetic.batch(10 * rank, 1, CFG["volume_size"], CFG["xray_img_size"])
    # local gradient without DDP
    loss = crit(model(xr), ct)["total_loss"]
    loss.backward()
    local = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    model.zero_grad(set_to_none=True)
    ddp = T.wrap_ddp(model)
    opt = torch.optim.AdamW(ddp.parameters(), lr=1e-3, weight_decay=0.01)
    # one harness step with an enormous clip (so the clip is the identity) to read the reduced gradients afterwards
    opt.zero_grad(set_to_none=True)
    with torch.autocast("cpu", enabled=False):
        ld = crit(ddp(xr), ct)
    ld["total_loss"].backward()
    reduced = {k: p.grad.clone() for k, p in model.named_parameters() if p.grad is not None}
    loss_dict = T.train_step(ddp, crit, opt, None, xr, ct, gradient_clip=1.0, autocast_device="cpu")
    torch.save({"local": local, "reduced": reduced, "params": {k: v.detach().clone() for k, v in model.state_dict().items()},
                "keys": sorted(loss_dict)}, os.path.join(out_dir, f"rank{rank}.pt"))
    T.cleanup_ddp()


@pytest.mark.timeout(600)
def test_two_rank_gloo_step(tmp_path):
    world, port = 2, 29517
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(world))
    assert r0["keys"] == ["l1_loss", "ssim_loss", "total_loss"]
    for k in r0["reduced"]:
        mean_local = (r0["local"][k] + r1["local"][k]) / 2
        assert torch.allclose(r0["reduced"][k], mean_local, rtol=1e-4, atol=1e-7), k
        assert torch.equal(r0["reduced"][k], r1["reduced"][k]), k
    for k in r0["params"]:
        assert torch.equal(r0["params"][k], r1["params"][k]), k
