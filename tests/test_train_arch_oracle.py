"""The torch restatement of unet / res_unet used by the train-step tests (oracle/train_ref.py:graph_loss_and_grads)
against the NumPy forward of the same graphs (oracle/: already pinned by the golden logits fixtures), and the Dropout
mask restatement's statistics."""
import numpy as np
import pytest


@pytest.mark.parametrize("arch,shape", [("unet", (32, 64)), ("res_unet", (40, 50))])
def test_torch_graph_matches_numpy_forward(oracle_mod, arch, shape):
    from oracle.train_ref import graph_loss_and_grads
    from pseg_amd import synth
    img, _, mask = synth.synth_page(4, max(shape[0], 96), max(shape[1], 96), 3)
    img, mask = np.ascontiguousarray(img[:shape[0], :shape[1]]), np.ascontiguousarray(mask[:shape[0], :shape[1]])
    Wt = oracle_mod.init_weights(arch, 3, seed=2, gain=1.2, bias_scale=0.05)
    loss, grads, z = graph_loss_and_grads(arch, Wt, img, mask)
    z_np = oracle_mod.forward(arch, Wt, img)
    assert z.shape == z_np.shape and np.abs(z - z_np).max() <= 1e-4 * max(1.0, np.abs(z_np).max())
    assert np.isfinite(loss) and list(grads) == list(Wt)
    assert all(np.abs(g).max() > 0 for g in grads.values())


def test_dropout_mask_restatement():
    from oracle.train_ref import dropout_keep, dropout_key
    k = dropout_keep(1 << 16, dropout_key(7, 0, 10), 0.5)
    assert 0.48 < k.mean() < 0.52
    assert not np.array_equal(k, dropout_keep(1 << 16, dropout_key(7, 1, 10), 0.5))
    assert not np.array_equal(k, dropout_keep(1 << 16, dropout_key(7, 0, 13), 0.5))
    assert 0.28 < dropout_keep(1 << 16, 12345, 0.7).mean() < 0.32
