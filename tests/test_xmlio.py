"""OpenCV-FileStorage depth XML (main.cpp:112-114,146-149): reader/writer round trip and a sample in
OpenCV 2.4's own number formatting."""
import numpy as np
import pytest

from kinectdepthmapenhancement_amd import xmlio

SAMPLE = """<?xml version="1.0"?>
<opencv_storage>
<averaged_depth type_id="opencv-matrix">
  <rows>2</rows>
  <cols>3</cols>
  <dt>f</dt>
  <data>
    0. 1234. 1.23456787e+03 8.15000000e+02 .Inf
    -.Inf</data></averaged_depth>
<depth type_id="opencv-matrix">
  <rows>2</rows>
  <cols>3</cols>
  <dt>f</dt>
  <data>
    1. 2. 3. 4.50000000e+00 .Nan 6.</data></depth>
</opencv_storage>
"""


def test_reads_opencv_formatting(tmp_path):
    p = tmp_path / "depth.xml"
    p.write_text(SAMPLE)
    depth, avg = xmlio.read_depth_xml(str(p))
    assert depth.dtype == np.float32 and depth.shape == (2, 3)
    assert np.array_equal(depth[0], [1, 2, 3]) and depth[1, 0] == 4.5 and np.isnan(depth[1, 1]) and depth[1, 2] == 6
    assert avg[0, 1] == 1234 and avg[0, 2] == np.float32(1234.56787) and avg[1, 1] == np.inf and avg[1, 2] == -np.inf


def test_round_trip_is_bit_exact(tmp_path, synth):
    _, depth, clean = synth.make_frame(3, 96, 64, clean=True)
    p = str(tmp_path / "d.xml")
    xmlio.write_depth_xml(p, depth, clean)
    d2, a2 = xmlio.read_depth_xml(p)
    assert np.array_equal(d2, depth) and np.array_equal(a2, clean)
    text = open(p).read()
    assert text.index("<averaged_depth") < text.index("<depth ")      # the reference writes averaged_depth first


def test_errors(tmp_path):
    p = tmp_path / "bad.xml"
    p.write_text("<opencv_storage><depth type_id=\"opencv-matrix\"><rows>1</rows><cols>2</cols><dt>f</dt><data>1.</data></depth></opencv_storage>")
    with pytest.raises(ValueError):
        xmlio.read_matrices(str(p))
    p.write_text("<notstorage/>")
    with pytest.raises(ValueError):
        xmlio.read_matrices(str(p))
