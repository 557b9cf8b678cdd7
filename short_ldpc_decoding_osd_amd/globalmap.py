"""Process-wide settings dictionary with the reference's ``globalmap`` surface
(LDPC_128/Ldpc_128_testing/globalmap.py:8-22; stage settings PB_OSD/globalmap.py:26-47,
FS_OSD/globalmap.py:28-50).  The decoders read the same keys the reference reads:
'code_parameters', 'num_iterations', 'selected_decoder_type', 'order_limit',
'termination_num_threshlod' (sic), 'd_min', 'tau_psc', 'pb_osd', 'fs_osd', 'convention_osd',
'miracle_view'.  Unlike the reference a missing key raises KeyError instead of printing.
"""
from __future__ import annotations

map = {}  # noqa: A001  (name kept from the reference)


def set_map(key, value):
    map[key] = value


def del_map(key):
    map.pop(key, None)


def get_map(key, default=KeyError):
    if key == "all":
        return map
    if key in map:
        return map[key]
    if default is KeyError:
        raise KeyError(f"globalmap: key '{key}' was never set")
    return default


def global_setting(argv, stage="PB"):
    """Positional settings of the OSD stages: ``prog snr_lo snr_hi snr_num unit_batch T Hfile type``
    (PB_OSD/globalmap.py:26-47, FS_OSD/globalmap.py:28-50) with the reference's defaults."""
    from .fill_matrix_info import Code

    set_map('snr_lo', float(argv[1]))
    set_map('snr_hi', float(argv[2]))
    set_map('snr_num', int(argv[3]))
    set_map('unit_batch_size', int(argv[4]))
    set_map('num_iterations', int(argv[5]))
    set_map('H_filename', argv[6])
    set_map('selected_decoder_type', argv[7])
    set_map('ALL_ZEROS_CODEWORD_TRAINING', False)
    set_map('code_parameters', Code(get_map('H_filename')))
    set_map('order_limit', 3)
    set_map('termination_num_threshlod', 100)
    set_map('miracle_view', False)
    set_map('convention_osd', False)
    if stage.upper() == "FS":
        set_map('fs_osd', True)
        set_map('d_min', 14)
        set_map('tau_psc', 30)
    else:
        set_map('pb_osd', True)
