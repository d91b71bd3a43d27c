"""The reference's own tests (tests/test_batching.py, tests/test_cvcl.py), adapted: same assertions -- batched forward ==
per-sample forward for the vision and text encoders (atol 1e-5 / 1e-4), and the shape smoke of the README usage -- on
random-init weights (the published checkpoint is not reachable without a network)."""
import argparse
import contextlib
import io
import itertools

import pytest
import torch

pytestmark = pytest.mark.gpu


def random_padded_tensor(dev, seed):                     # reference tests/test_batching.py:45-57
    g = torch.Generator().manual_seed(seed)
    x = torch.zeros((4, 16), dtype=torch.long)
    x_len = torch.randint(low=1, high=16, size=(4,), generator=g)
    for i in range(len(x)):
        x[i, :x_len[i]] = torch.randint(low=1, high=16, size=(int(x_len[i]),), generator=g)
    return x.to(dev), x_len.to(dev)


def _text_encoder(dev, kind, embedding_type):
    from multimodal.multimodal import TextEncoder
    vocab = {f"w{i}": i for i in range(10000)}
    args = argparse.Namespace(text_encoder=kind, embedding_type=embedding_type, embedding_dim=128, crange=1, dropout_i=0.0,
                              dropout_o=0.0, pos_embed_type="no_pos_embed", captioning=False, attention=False, attention_gate=False)
    with contextlib.redirect_stdout(io.StringIO()):
        return TextEncoder(vocab, 2048, args).to(dev).eval()


def test_cnn(dev):                                       # reference test_cnn (:21-42); random init instead of pretrained
    from multimodal.multimodal import VisionEncoder
    args = argparse.Namespace(embedding_dim=128, pretrained_cnn=False, finetune_cnn=False, embedding_type="flat",
                              cnn_model="resnext50_32x4d", cnn_dino=False, vit_dino=False)
    with contextlib.redirect_stdout(io.StringIO()):
        model = VisionEncoder(args).to(dev).eval()
    x = torch.rand([4, 3, 224, 224], device=dev)
    with torch.no_grad():
        y_batched, fmap = model(x)
        y_unbatched = model._forward_unbatched(x)
    assert y_batched.shape == (4, 128) and fmap.shape == (4, 2048, 7, 7)
    assert torch.allclose(y_batched, y_unbatched, atol=1e-5)


@pytest.mark.parametrize("embedding_type", ["spatial", "flat"])
def test_embedding(dev, embedding_type):                 # reference test_spatial_embedding / test_flat_embedding
    model = _text_encoder(dev, "embedding", embedding_type)
    x, x_len = random_padded_tensor(dev, 1)
    with torch.no_grad():
        y_batched = model(x, x_len)[0]
        y_unbatched = model._forward_unbatched(x, x_len)
    assert y_batched.shape == ((4, 16, 128) if embedding_type == "spatial" else (4, 128))
    assert torch.allclose(y_batched, y_unbatched, atol=1e-5)


@pytest.mark.parametrize("embedding_type,bidirectional", list(itertools.product(["flat", "spatial"], [True, False])))
def test_lstm(dev, embedding_type, bidirectional):       # reference test_lstm: the cartesian product of its configs
    model = _text_encoder(dev, "bilstm" if bidirectional else "lstm", embedding_type)
    x, x_len = random_padded_tensor(dev, 2)
    with torch.no_grad():
        y_batched = model(x, x_len)[0]
        y_unbatched = model._forward_unbatched(x, x_len)
    if embedding_type == "spatial":                      # batched output is trimmed to the longest sequence
        y_unbatched = y_unbatched[:, :y_batched.shape[1]]
    assert torch.allclose(y_batched, y_unbatched, atol=1e-4)


def test_cvcl_usage_smoke(dev):                          # reference tests/test_cvcl.py / README usage, random-init model
    import train
    argv = ("--dataset synthetic --gpus 1 --text_encoder embedding --embedding_dim 512 --normalize_features --fix_temperature "
            "--lambda_lm 0 --optimize_unused --fast_dev_run --checkpoint_callback False --logger False --batch_size 4").split()
    with contextlib.redirect_stdout(io.StringIO()):
        _trainer, cvcl = train.main(argv)
    cvcl = cvcl.to(dev).eval()
    images = torch.rand(4, 3, 224, 224, device=dev)
    with torch.no_grad():
        image_features = cvcl.encode_image(images)
        texts, texts_len = cvcl.tokenize(["ball"])
        texts, texts_len = texts.to(dev), texts_len.to(dev)
        texts_features = cvcl.encode_text(texts, texts_len)
        logits_per_image, logits_per_text = cvcl(images, texts, texts_len)
    assert image_features.shape == (4, 512) and texts_features.shape == (1, 512)
    assert logits_per_image.shape == (4, 1) and logits_per_text.shape == (1, 4)
