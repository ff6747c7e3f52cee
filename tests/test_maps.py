import os

import numpy as np

from conftest import GOLDEN


def test_trinary_rule(maps_mod):
    img = np.array([[0, 100, 200, 255], [254, 210, 205, 50]], np.uint8)
    g = maps_mod.trinary_from_image(img, False, 0.65, 0.196)
    # occ = (255-shade)/255: 0->1.0 occ; 100->0.608 unknown; 200->0.216 unknown; 255->0 free
    # row flip: image row 0 is the TOP
    assert g.tolist() == [[0, 0, -1, 100], [100, -1, -1, 0]]
    rgba = np.zeros((1, 2, 4), np.uint8)
    rgba[0, 0] = (255, 255, 255, 0); rgba[0, 1] = (0, 0, 0, 255)
    assert maps_mod.trinary_from_image(rgba, False, 0.65, 0.196).tolist() == [[0, 100]]
    assert maps_mod.trinary_from_image(rgba, True, 0.65, 0.196).tolist() == [[100, 0]]


def test_fixture_cell_counts_match_survey_appendix_c(spielberg, sibal1, maps_mod):
    d = spielberg.data
    assert d.shape == (2000, 2000)
    assert ((d > 50).sum(), (d == 0).sum(), (d < 0).sum()) == (33998, 3960078, 5924)
    d = sibal1.data
    assert d.shape == (177, 350)
    assert ((d > 50).sum(), (d == 0).sum(), (d < 0).sum()) == (35002, 26948, 0)
    assert float(spielberg.resolution) == 0.057959999889135361
    ic = maps_mod.load_npz(os.path.join(GOLDEN, "map_icra_2_clean.npz"))
    assert ((ic.data > 50).sum(), (ic.data == 0).sum()) == (108449, 46759)


def test_npz_roundtrip(maps_mod, sibal1, tmp_path):
    p = str(tmp_path / "m.npz")
    maps_mod.save_npz(sibal1, p)
    m2 = maps_mod.load_npz(p)
    assert np.array_equal(m2.data, sibal1.data) and m2.resolution == sibal1.resolution
    assert (m2.origin_x, m2.origin_y) == (sibal1.origin_x, sibal1.origin_y)


def test_synthetic_levine(maps_mod):
    m = maps_mod.synthetic_levine()
    assert m.data.shape == (2049, 2049) and float(m.resolution) == 0.05000000074505806
    assert (m.origin_x, m.origin_y) == (-51.224998, -51.224998)
    assert (m.data == 0).sum() > 200000 and (m.data > 50).sum() > 5000
    m2 = maps_mod.synthetic_levine()
    assert np.array_equal(m.data, m2.data)


def test_scale_and_raw_modes(maps_mod):
    img = np.array([[0, 100, 200, 255]], np.uint8)            # occ = 1.0, 0.608, 0.216, 0.0
    g = maps_mod.grid_from_image(img, "scale", False, 0.65, 0.196)
    assert g.tolist() == [[100, int(np.rint(99 * (155 / 255 - 0.196) / (0.65 - 0.196))), int(np.rint(99 * (55 / 255 - 0.196) / (0.65 - 0.196))), 0]]
    la = np.zeros((1, 2, 2), np.uint8); la[0, 0] = (0, 255); la[0, 1] = (0, 10)     # grey + alpha
    assert maps_mod.grid_from_image(la, "scale").tolist() == [[100, -1]]
    g = maps_mod.grid_from_image(img, "raw")
    assert g.tolist() == [[-1, -1, 55, 0]]
    g = maps_mod.grid_from_image(np.array([[155, 205, 255]], np.uint8), "raw")
    assert g.tolist() == [[100, 50, 0]]


def test_yaml_loader_png_pgm_modes(maps_mod, tmp_path):
    """PNG and PGM through the YAML loader: trinary default, negate, explicit mode, relative image path, flip."""
    from PIL import Image
    rng = np.random.default_rng(3)
    shade = rng.choice(np.array([0, 120, 254, 255], np.uint8), size=(7, 9))
    Image.fromarray(shade).save(tmp_path / "m.png")
    Image.fromarray(shade).save(tmp_path / "m.pgm")
    rgb = np.stack([shade, shade, shade], axis=2)
    Image.fromarray(rgb).save(tmp_path / "c.png")
    for image, extra in (("m.png", ""), ("m.pgm", ""), ("c.png", ""), ("m.png", "negate: 1\n"), ("m.pgm", "mode: scale\n")):
        y = tmp_path / "map.yaml"
        y.write_text(f"image: {image}\nresolution: 0.05\norigin: [-1.5, 2.0, 0.0]\noccupied_thresh: 0.65\nfree_thresh: 0.196\n{extra}")
        m = maps_mod.load_map_yaml(str(y))
        mode = "scale" if "scale" in extra else "trinary"
        want = maps_mod.grid_from_image(shade, mode, "negate" in extra, 0.65, 0.196)
        assert np.array_equal(m.data, want), (image, extra)
        assert m.data.shape == (7, 9) and m.resolution == np.float32(0.05) and (m.origin_x, m.origin_y) == (-1.5, 2.0)
    # the bottom image row is grid row 0
    m = maps_mod.load_map_yaml(str(y))
    assert np.array_equal(m.data[0] > 50, maps_mod.grid_from_image(shade, "scale")[0] > 50)
