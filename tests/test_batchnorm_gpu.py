"""BatchNormalization at res_unet's bn_act sites (lib/model.py:265-271; SURVEY 8 row a7: "conv_block_simple / BN
placement"), PSEG_FLAG_BATCHNORM / Engine(batch_norm=True).

Inference (moving statistics): float32 engine bit-identical to the NumPy restatement (oracle/models.py:_Ctx.bn), bf16
engine within the unet / res_unet logit bar, label-exact mode equal to the float32 engine.
Training (batch statistics): loss within 1e-4 relative and every gradient within 2e-3 of its scale against torch
autograd in float64 (oracle/train_ref.py), moving statistics updated as the fused Keras kernel does, evaluation on the
moving statistics.  TensorFlow is absent here and the reference never switches the layer on: parity unpinned against
Keras itself."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _sample(seed, H, W, C):
    from pseg_amd import synth
    img, _, mask = synth.synth_page(seed, max(H, 96), max(W, 96), C)
    return np.ascontiguousarray(img[:H, :W]), np.ascontiguousarray(mask[:H, :W])


def test_weight_table_is_in_keras_creation_order(gpu, oracle_mod):
    eng = gpu.Engine("res_unet", 3, mode=gpu.MODE_F32_EXACT, batch_norm=True)
    want = []
    for name, kind, shp, _ in oracle_mod.models.weight_specs("res_unet", 3, batch_norm=True):
        if kind == "bn":
            want += [(name + s, tuple(shp)) for s in ("/gamma", "/beta", "/moving_mean", "/moving_variance")]
        else:
            want += [(name + "/kernel", tuple(shp)), (name + "/bias", (shp[3] if kind == "conv" else shp[2],))]
    got = [(n, tuple(s)) for n, s in eng.weight_specs()]
    assert got == want
    assert sum(1 for n, _ in got if n.endswith("/gamma")) == 28      # 2 stem + 3 x 8 residual blocks + 2 bridge
    assert dict(got)["batch_normalization_16/gamma"] == (768,)       # first decoder block: BN over concat [up(512), skip(256)]
    eng.close()
    with pytest.raises(gpu.PsegError, match="BATCHNORM"):
        gpu.Engine("fcn_skip", 3, batch_norm=True)


@pytest.mark.parametrize("C,shape", [(3, (64, 96)), (4, (70, 50))])
def test_predict_float32_is_bit_identical_to_the_oracle(gpu, oracle_mod, C, shape):
    Wt = oracle_mod.init_weights("res_unet", C, seed=42, gain=1.5, bias_scale=0.05, batch_norm=True)
    img, _ = _sample(3, shape[0], shape[1], C)
    lo, po, pr = oracle_mod.predict_single_data("res_unet", Wt, img, "f32")
    eng = gpu.Engine("res_unet", C, mode=gpu.MODE_F32_EXACT, batch_norm=True)
    eng.set_weights(Wt)
    z, p, l = eng.predict(img)
    assert np.array_equal(z, lo) and np.array_equal(l, pr)
    # and it is not the graph without the layers
    e0 = gpu.Engine("res_unet", C, mode=gpu.MODE_F32_EXACT)
    e0.set_weights({k: v for k, v in Wt.items() if "batch_normalization" not in k})
    assert not np.array_equal(e0.predict(img)[0], z)
    e0.close()
    eng.close()


def test_predict_bf16_and_label_exact(gpu, oracle_mod):
    import torch
    C, (H, W) = 3, (96, 64)
    Wt = oracle_mod.init_weights("res_unet", C, seed=7, gain=1.5, bias_scale=0.05, batch_norm=True)
    img, _ = _sample(5, H, W, C)
    lo = oracle_mod.forward("res_unet", Wt, img, "f32")
    eb = gpu.Engine("res_unet", C, mode=gpu.MODE_BF16, batch_norm=True)
    eb.set_weights(Wt)
    z = eb.predict(img)[0]
    assert np.abs(z - lo).max() <= 0.03 * max(1.0, float(np.abs(lo).max()))
    dev = torch.device("cuda:0")
    d_img = torch.from_numpy(img).to(dev)
    lab = torch.empty((H, W), dtype=torch.uint8, device=dev)
    eb.predict_exact_labels_device(d_img.data_ptr(), H, W, lab.data_ptr(), stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize()
    assert np.array_equal(lab.cpu().numpy(), np.argmax(lo, -1).astype(np.uint8))
    eb.close()


def _compare(g, g_o, tol=2e-3):
    assert list(g.keys()) == list(g_o.keys())
    for k in g_o:
        if "moving_" in k:
            assert not np.any(g[k]), k        # not trained
            continue
        scale = np.abs(g_o[k]).max() + 1e-12
        err = np.abs(g[k] - g_o[k]).max()
        if scale < 1e-6:
            # a per-channel constant directly in front of a batch-statistics normalisation (the bias of the stem conv, of
            # conv_block 1 and of the shortcuts; e5's shortcut beta, read by nothing but the bridge) has a zero gradient --
            # the mean removes it; what is left is float32 summation noise of the activation gradients
            assert k.endswith(("/bias", "/beta")) and err < 5e-5, (k, err, scale)
            continue
        assert err <= tol * scale + 1e-9, "%s: max err %g vs scale %g" % (k, err, scale)


@pytest.mark.parametrize("C,shape", [(3, (64, 96)), (4, (40, 50))])
def test_train_step_matches_autograd(gpu, oracle_mod, C, shape):
    from oracle.train_ref import graph_loss_and_grads
    from oracle.models import BN_MOMENTUM
    Wt = oracle_mod.init_weights("res_unet", C, seed=11, gain=1.2, bias_scale=0.05, batch_norm=True)
    img, mask = _sample(2, shape[0], shape[1], C)
    stats = {}
    loss_o, g_o, _ = graph_loss_and_grads("res_unet", Wt, img, mask, float64=True, bn_stats=stats)
    eng = gpu.Engine("res_unet", C, mode=gpu.MODE_F32_EXACT, batch_norm=True)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    loss = eng.train_forward_backward(img, mask)[0]
    assert abs(loss - loss_o) <= 1e-4 * abs(loss_o), (loss, loss_o)
    _compare(eng.gradients(), g_o)
    # moving statistics after one training forward (nothing else moved: no apply yet)
    after = eng.get_weights()
    assert len(stats) == 28
    dec = 1.0 - float(BN_MOMENTUM)
    for name, (mean, var, n) in stats.items():
        mm, mv = Wt[name + "/moving_mean"], Wt[name + "/moving_variance"]
        want_m = mm - (mm - mean) * dec
        want_v = mv - (mv - var * (n / (n - 1.0))) * dec
        assert np.allclose(after[name + "/moving_mean"], want_m, rtol=1e-5, atol=1e-6), name
        assert np.allclose(after[name + "/moving_variance"], want_v, rtol=1e-5, atol=1e-6), name
        assert np.array_equal(after[name + "/gamma"], Wt[name + "/gamma"])
    # evaluation (Keras validation / predict) normalises with the moving statistics: the loss is the cross-entropy of
    # the inference logits under the updated table
    z = oracle_mod.forward("res_unet", after, img, "f32").astype(np.float64)
    z -= z.max(-1, keepdims=True)
    ce = float(np.mean(np.log(np.exp(z).sum(-1)) - np.take_along_axis(z, mask[..., None].astype(np.int64), -1)[..., 0]))
    assert eng.eval_step(img, mask)[0] == pytest.approx(ce, rel=1e-4)
    assert np.array_equal(eng.get_weights()["batch_normalization/moving_mean"], after["batch_normalization/moving_mean"])   # eval moves nothing
    eng.close()


def test_training_reduces_the_loss_and_the_result_predicts(gpu, oracle_mod):
    Wt = oracle_mod.init_weights("res_unet", 3, seed=5, gain=1.0, bias_scale=0.0, batch_norm=True)
    pages = [_sample(s, 64, 64, 3) for s in (0, 1)]
    eng = gpu.Engine("res_unet", 3, mode=gpu.MODE_F32_EXACT, batch_norm=True)
    eng.set_weights(Wt)
    eng.train_init(clipnorm=1.0)
    first = np.mean([eng.train_forward_backward(*p)[0] for p in pages])
    for step in range(80):
        eng.train_forward_backward(*pages[step % 2])
        eng.train_apply(1e-3)
    last = np.mean([eng.train_forward_backward(*p)[0] for p in pages])
    assert np.isfinite(last) and last < 0.5 * first, (first, last)
    Wn = eng.get_weights()
    assert not np.array_equal(Wn["batch_normalization_3/gamma"], Wt["batch_normalization_3/gamma"])
    assert not np.array_equal(Wn["batch_normalization_3/moving_variance"], Wt["batch_normalization_3/moving_variance"])
    # trained table through the float32 predict path == the oracle; the bf16 engine accepts it
    lo = oracle_mod.forward("res_unet", Wn, pages[0][0], "f32")
    assert np.array_equal(eng.predict(pages[0][0])[0], lo)
    eb = gpu.Engine("res_unet", 3, mode=gpu.MODE_BF16, batch_norm=True)
    eb.set_weights(Wn)
    assert eb.predict(pages[0][0], want_logits=False, want_probs=False)[2].shape == (64, 64)
    eb.close()
    eng.close()
