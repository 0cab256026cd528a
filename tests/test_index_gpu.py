"""The table `image` around the hot path (search.rs:13-18, clip.rs:42-151) on a real MI355X:
path column, dedupe-by-path, the refine + kNN search flow, persistence, and the scan loop."""
import os

import numpy as np
import pytest

from image_search_amd import synth
from image_search_amd.clip import PRECISION_F32, Model
from image_search_amd.search import ImageIndex, embed_all_images_in_dir, refine_query

pytestmark = pytest.mark.gpu


def _index(n=300, dim=768, media="/data/pics/"):
    ix = ImageIndex(dim, 0, media)
    emb = synth.gen_f32(3, 0, n * dim).reshape(n, dim)
    paths = [f"{media}a/{i:04d}.jpg" for i in range(n)]
    ix.insert(paths[:100], emb[:100])
    ix.insert(paths[100:], emb[100:])
    return ix, paths, emb


def test_existing_insert_and_lookup(built):
    ix, paths, emb = _index()
    assert len(ix) == 300 and ix.existing([paths[3], "/nope.jpg", paths[250]]) == {paths[3], paths[250]}
    rows, vecs = ix.embeddings_of([paths[250], paths[3], "/nope.jpg"])      # table order, not request order
    assert rows == [3, 250] and np.array_equal(vecs[0], emb[3]) and np.array_equal(vecs[1], emb[250])


def test_web_search_text_is_refine_then_knn(built, orc):
    from oracle.binding import orc_knn
    ix, paths, emb = _index()
    text = synth.gen_f32(8, 0, 768)
    plain = ix.web_search_text(text, [], k=10)
    ids, dist = orc_knn(orc, text, emb, 10)
    assert [r[0] for r in plain] == [int(i) for i in ids]
    assert np.array_equal(np.array([r[2] for r in plain], np.float32), dist)
    assert plain[0][1] == f"media/a/{int(ids[0]):04d}.jpg"                       # search.rs:104-109 path mapping
    marked = ["media/a/0007.jpg", "media/a/0123.jpg", "other/x.jpg", "media/missing.jpg"]
    got = ix.web_search_text(text, marked, k=10)
    q = refine_query(text, [emb[7], emb[123]])
    ids2, dist2 = orc_knn(orc, q, emb, 10)
    assert [r[0] for r in got] == [int(i) for i in ids2]
    assert ix.web_search_text(text, ["media/missing.jpg"], k=10) == plain          # nothing found: the text vector
    assert len(ix.web_search_text(text, [], k=1000)) == 300                         # K above the table size


def test_save_load_roundtrip(built, tmp_path):
    ix, paths, emb = _index(n=1000)
    ix.save(str(tmp_path / "ix"))
    again = ImageIndex.load(str(tmp_path / "ix"))
    assert again.paths == paths and again.media_dir == ix.media_dir
    assert np.array_equal(again.table.rows(0, 1000), emb)
    t = synth.gen_f32(5, 0, 768)
    assert again.web_search_text(t, ["media/a/0001.jpg"], 25) == ix.web_search_text(t, ["media/a/0001.jpg"], 25)
    with open(tmp_path / "ix" / "embedding.miknn", "r+b") as f:
        f.truncate(4096)
    with pytest.raises(Exception):
        ImageIndex.load(str(tmp_path / "ix"))


def test_scan_loop_embeds_new_files_once(built, orc, tmp_path):
    from PIL import Image
    from oracle.binding import orc_resize_catmullrom
    cfg = synth.VitConfig.tiny()
    wpath = str(tmp_path / "tiny.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 1), wpath, {"num_attention_heads": cfg.heads})
    m = Model.from_file(wpath, 0, PRECISION_F32)
    media = tmp_path / "media"
    (media / "sub").mkdir(parents=True)
    imgs = {}
    for i, (h, w) in enumerate([(90, 120), (56, 56), (300, 200), (64, 33), (77, 200)]):
        p = media / ("sub" if i % 2 else "") / f"im{i}.png"
        imgs[str(p)] = synth.photo_u8(40 + i, h, w)
        Image.fromarray(imgs[str(p)]).save(p)
    (media / "notes.txt").write_text("not an image")
    (media / "broken.jpg").write_bytes(b"\xff\xd8 not really a jpeg")
    ix = ImageIndex(cfg.proj, 0, str(media) + "/")
    assert embed_all_images_in_dir(m, ix, str(media), image_chunk_size=2, shuffle_seed=1) == 5
    assert sorted(ix.paths) == sorted(imgs)
    assert embed_all_images_in_dir(m, ix, str(media), image_chunk_size=2) == 0        # idempotent (clip.rs:74-87)
    for row, p in enumerate(ix.paths):                                                  # each row is its own file's embedding
        px = synth.preprocess_rgb8(orc_resize_catmullrom(orc, imgs[p], m.image, m.image)[None])
        assert np.array_equal(ix.table.rows(row, 1), m.forward(px))


def test_scan_loop_survives_a_symlink_cycle_and_the_index_a_crash_between_its_two_files(built, tmp_path):
    from PIL import Image
    cfg = synth.VitConfig.tiny()
    wpath = str(tmp_path / "tiny.safetensors")
    synth.save_safetensors(synth.vit_weights(cfg, 1), wpath, {"num_attention_heads": cfg.heads})
    m = Model.from_file(wpath, 0, PRECISION_F32)
    media = tmp_path / "media"
    (media / "a" / "b").mkdir(parents=True)
    for i, d in enumerate(("", "a", "a/b")):
        Image.fromarray(synth.photo_u8(70 + i, 50, 60)).save(media / d / f"f{i}.png")
    os.symlink(media, media / "a" / "b" / "loop")          # WalkDir follows links AND detects cycles; so must the walk here
    os.symlink(media / "a", media / "again")               # the same directory under a second name: its files once
    ix = ImageIndex(cfg.proj, 0, str(media) + "/")
    assert embed_all_images_in_dir(m, ix, str(media), image_chunk_size=2, shuffle_seed=2) == 3
    d = str(tmp_path / "ix")
    ix.save(d)
    # a crash after the embedding file of a LATER save was renamed, before its path file was: newer embeddings, older paths
    ix.insert([str(media / "extra.png")], np.ones((1, cfg.proj), np.float32))
    from image_search_amd._lib import check, lib
    check(lib().mi_knn_save(ix.table._h, os.path.join(d, "embedding.miknn").encode()))
    back = ImageIndex.load(d, 0, cfg.proj)
    assert len(back) == 3 and len(back.table) == 3 and sorted(back.paths) == sorted(ix.paths[:3])
    assert [r[:2] for r in back.web_search_text(np.ones(cfg.proj, np.float32), [], k=3)] == \
           [r[:2] for r in ix.web_search_text(np.ones(cfg.proj, np.float32), [], k=4) if r[0] < 3][:3]
    assert not os.path.exists(os.path.join(d, "embedding.miknn.tmp")) and not os.path.exists(os.path.join(d, "image_path.bin.tmp"))
    m.close()


def test_end_to_end_text_query_over_scanned_directory(built, tmp_path):
    """The reference's whole request path on the device: scan a media directory (decode -> resize ->
    tower -> rows), then `web_search_text`: text tower -> refine with a marked image -> kNN -> paths.
    Checked against the same flow assembled from the parts that are individually pinned."""
    from PIL import Image
    from image_search_amd.clip import TextModel
    vcfg, tcfg = synth.VitConfig.tiny(), synth.TextConfig.tiny()
    assert vcfg.proj == tcfg.proj
    vpath, tpath = str(tmp_path / "v.safetensors"), str(tmp_path / "t.safetensors")
    synth.save_safetensors(synth.vit_weights(vcfg, 1), vpath, {"num_attention_heads": vcfg.heads})
    synth.save_safetensors(synth.vit_weights(tcfg, 2), tpath, {"num_attention_heads": tcfg.heads})
    vm, tm = Model.from_file(vpath, 0, PRECISION_F32), TextModel.from_file(tpath)
    media = tmp_path / "media"
    media.mkdir()
    for i in range(12):
        Image.fromarray(synth.photo_u8(60 + i, 40 + 7 * i, 90 - 3 * i)).save(media / f"p{i:02d}.png")
    ix = ImageIndex(vcfg.proj, 0, str(media) + "/")
    assert embed_all_images_in_dir(vm, ix, str(media), image_chunk_size=5, shuffle_seed=3) == 12
    text = tm.embed(synth.token_ids(tcfg, 5, 1))[0]
    marked = "media/p03.png"
    got = ix.web_search_text(text, [marked], k=5)
    row = ix.paths.index(str(media / "p03.png"))
    q = refine_query(text, [ix.table.rows(row, 1)[0]])
    ids, dist = ix.table.knn(q, 5)
    assert [g[0] for g in got] == [int(i) for i in ids]
    assert all(g[1].startswith("media/p") for g in got)
    assert np.array_equal(np.array([g[2] for g in got], np.float32), dist)
