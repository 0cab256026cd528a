"""image_prepare_resnet (server/src/clip.rs:153-175) on a real MI355X through the C ABI:
the CatmullRom resize kernels and the fused normalisation are BIT-EXACT against the CPU oracle
(byte outputs, and the f32 outputs of the same op order), for every shape class the `image`
crate's resampler distinguishes: down, up, mixed, one-pixel extents, equal sizes, extreme ratios."""
import numpy as np
import pytest

from image_search_amd import synth
from image_search_amd._lib import MiError
from image_search_amd.clip import PRECISION_F32, Model, image_prepare_resnet, resize_exact
from oracle.binding import orc_image_prepare_resnet, orc_resize_catmullrom

pytestmark = pytest.mark.gpu

SHAPES = [(300, 401), (37, 53), (1000, 61), (224, 224), (225, 223), (1, 1), (1, 500), (500, 1), (2, 3),
          (223, 224), (480, 640), (1080, 1920), (3000, 4000), (7000, 300)]


@pytest.mark.parametrize("hw", SHAPES)
def test_resize_bit_exact(built, orc, hw):
    img = synth.photo_u8(sum(hw), *hw)
    assert np.array_equal(resize_exact(img, 224, 224), orc_resize_catmullrom(orc, img, 224, 224))


def test_resize_other_targets_and_golden(built, orc):
    import os
    from conftest import GOLDEN
    img = synth.photo_u8(1, 199, 333)
    for (nw, nh) in ((224, 224), (64, 48), (500, 777), (333, 199), (2, 1), (2, 224)):
        assert np.array_equal(resize_exact(img, nw, nh), orc_resize_catmullrom(orc, img, nw, nh)), (nw, nh)
    g = np.load(os.path.join(GOLDEN, "resize.npz"))
    for name in ("down", "up", "mixed"):
        h, w = (int(v) for v in g[f"{name}_hw"])
        assert np.array_equal(resize_exact(synth.photo_u8(int(g[f"{name}_seed"]), h, w), 224, 224), g[f"{name}_oracle"])


@pytest.mark.parametrize("hw", [(300, 401), (37, 53), (224, 224), (3000, 4000)])
def test_image_prepare_resnet_bit_exact(built, orc, hw):
    img = synth.photo_u8(11 + hw[0], *hw)
    got = image_prepare_resnet(img)
    ref = orc_image_prepare_resnet(orc, img)
    assert got.dtype == np.float32 and got.shape == (3, 224, 224)
    assert np.array_equal(got.view(np.uint32), ref.view(np.uint32))


def test_limits_are_codes(built):
    with pytest.raises(MiError) as e:
        resize_exact(np.zeros((0, 5, 3), np.uint8), 224, 224)
    assert e.value.code == -5
    with pytest.raises(MiError) as e:
        resize_exact(np.zeros((2, 40000, 3), np.uint8), 224, 224)   # extent above 32768
    assert e.value.code == -5
    with pytest.raises(MiError) as e:
        resize_exact(np.zeros((300, 8, 3), np.uint8), 4, 1)         # 300x reduction: more taps than the LDS table
    assert e.value.code == -5


def test_embed_images_equals_two_step_flow(built, orc, tmp_path):
    """mi_clip_embed_images == mi_clip_embed(image_prepare_resnet(img) for img in chunk): same
    device kernels after the resize, so the embeddings are identical bits; images of different
    sizes in one chunk, more images than one internal pass."""
    cfg = synth.VitConfig.tiny()
    path = str(tmp_path / "tiny.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 1), path, {"num_attention_heads": cfg.heads})
    m = Model.from_file(path, 0, PRECISION_F32)
    sizes = [(300, 401), (m.image, m.image), (37, 53), (640, 480), (1000, 61)]
    imgs = [synth.photo_u8(20 + i, *hw) for i, hw in enumerate(sizes)]
    two_step = m.forward(synth.preprocess_rgb8(np.stack([orc_resize_catmullrom(orc, im, m.image, m.image) for im in imgs])))
    fused = m.forward_images(imgs)
    assert np.array_equal(fused.view(np.uint32), two_step.view(np.uint32))
    assert m.forward_images([]).shape == (0, m.proj)
