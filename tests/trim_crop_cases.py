"""The trim / crop cases of the reference's test-suite (xrspatial/tests/test_zonal.py:1047-1211), as data:
(function, input raster, values / zones_ids, expected window).  Used to pin the oracle (test_oracle_golden.py) and to
check the device path (test_gpu_parity.py)."""
import numpy as np

_A = np.array
I64 = np.int64

CASES = [
    # test_trim (:1047-1062)
    ("trim", _A([[0, 0, 0, 0], [0, 4, 0, 0], [0, 4, 4, 0], [0, 1, 1, 0], [0, 0, 0, 0]], dtype=I64), (0,),
     _A([[4, 0], [4, 4], [1, 1]], dtype=I64)),
    # test_trim_left_top (:1065-1082)
    ("trim", _A([[0, 0, 0, 0], [0, 4, 0, 3], [0, 4, 4, 3], [0, 1, 1, 3], [0, 1, 1, 3]], dtype=I64), (0,),
     _A([[4, 0, 3], [4, 4, 3], [1, 1, 3], [1, 1, 3]], dtype=I64)),
    # test_trim_right_top (:1085-1102)
    ("trim", _A([[0, 0, 0, 0], [4, 0, 3, 0], [4, 4, 3, 0], [1, 1, 3, 0], [1, 1, 3, 0]], dtype=I64), (0,),
     _A([[4, 0, 3], [4, 4, 3], [1, 1, 3], [1, 1, 3]], dtype=I64)),
    # test_trim_left_bottom (:1105-1122)
    ("trim", _A([[4, 0, 3, 0], [4, 4, 3, 0], [1, 1, 3, 0], [1, 1, 3, 0], [0, 0, 0, 0]], dtype=I64), (0,),
     _A([[4, 0, 3], [4, 4, 3], [1, 1, 3], [1, 1, 3]], dtype=I64)),
    # test_trim_right_bottom (:1125-1142)
    ("trim", _A([[0, 4, 0, 3], [0, 4, 4, 3], [0, 1, 1, 3], [0, 1, 1, 3], [0, 0, 0, 0]], dtype=I64), (0,),
     _A([[4, 0, 3], [4, 4, 3], [1, 1, 3], [1, 1, 3]], dtype=I64)),
    # test_crop (:1145-1162)
    ("crop", _A([[0, 4, 0, 3], [0, 4, 4, 3], [0, 1, 1, 3], [0, 1, 1, 3], [0, 0, 0, 0]], dtype=I64), (1, 3),
     _A([[4, 0, 3], [4, 4, 3], [1, 1, 3], [1, 1, 3]], dtype=I64)),
    # test_crop_nothing_to_crop (:1203-1213)
    ("crop", _A([[0, 4, 0, 3], [0, 4, 4, 3], [0, 1, 1, 3], [0, 1, 1, 3], [0, 0, 0, 0]], dtype=I64), (0,),
     _A([[0, 4, 0, 3], [0, 4, 4, 3], [0, 1, 1, 3], [0, 1, 1, 3], [0, 0, 0, 0]], dtype=I64)),
]
