"""The Python layout tables (assistive_gym_amd/model/compiler.py) mirror include/agx_blob.h."""
import os
import re

from assistive_gym_amd.model import compiler as L

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse_enums():
    src = open(os.path.join(ROOT, 'include', 'agx_blob.h')).read()
    src = re.sub(r'/\*.*?\*/', '', src, flags=re.S)
    vals = {}
    for body in re.findall(r'enum\s*\{(.*?)\}', src, flags=re.S):
        cur = -1
        for item in body.split(','):
            item = item.strip()
            if not item:
                continue
            if '=' in item:
                name, v = [x.strip() for x in item.split('=')]
                cur = int(v, 0)
            else:
                name, cur = item, cur + 1
            vals[name] = cur
    for name, v in re.findall(r'#define\s+(AGX_\w+)\s+\(?(-?\w+)\)?', src):
        try:
            vals[name] = int(v, 0)
        except ValueError:
            pass
    return vals


def test_layout_tables_match_header():
    v = parse_enums()
    for prefix, table in (('AGX_H_', L.H), ('AGX_P_', L.P), ('AGX_R_', L.R), ('AGX_F_', L.F), ('AGX_C_', L.C), ('AGX_G_', L.G),
                          ('AGX_T_', L.T), ('AGX_E_', L.E), ('AGX_X_', L.X_), ('AGX_XJ_', L.XJ), ('AGX_CL_', L.CL), ('AGX_CP_', L.CP), ('AGX_DR_', L.DR), ('AGX_AM_', L.AM)):
        for k, val in table.items():
            assert v[prefix + k] == val, (prefix + k, v[prefix + k], val)
    assert v['AGX_BLOB_MAGIC'] == L.MAGIC and v['AGX_BLOB_VERSION'] == L.VERSION
    assert v['AGX_BODY_ROBOT_BASE'] == L.BODY_ROBOT_BASE and v['AGX_BODY_FREE0'] == L.BODY_FREE0 and v['AGX_BODY_HUMAN0'] == L.BODY_HUMAN0
    for k, val in L.TAG.items():
        assert v['AGX_TAG_' + k] == val
    assert (v['AGX_CLOTH_MAX_COLORS'], v['AGX_CLOTH_THREADS'], v['AGX_CLOTH_NODE_CONTACTS']) == (L.CLOTH_MAX_COLORS, L.CLOTH_THREADS, L.CLOTH_NODE_CONTACTS)
    for k, val in L.KIND.items():
        assert v['AGX_KIND_' + k] == val


def test_blob_header(blob):
    assert blob.ndof == 14 and blob.nrobot == 10 and blob.nhdof == 4 and blob.nfree == 10 and blob.nfood == 8 and blob.act_dim == 7 and blob.obs_dim == 25
    assert blob.h['NWORDS'] == len(blob.words)
    assert blob.state_words == 3 * blob.ndof + 13 * blob.nfree + 7 + 7 * blob.nhuman + 2 * blob.nhdof + L.E['COUNT']
