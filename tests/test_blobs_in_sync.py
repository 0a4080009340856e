"""The committed model blobs are what the model compiler produces from the reference's assets today (a blob that was not recompiled after
a layout or compiler change would silently feed stale constants to the kernels).  Needs the reference's assets: skipped where
/root/reference is absent (the GPU box)."""
import os

import numpy as np
import pytest

from conftest import full

ASSETS = '/root/reference/assistive_gym/envs/assets'
LEAN = {'feeding_jaco', 'bed_bathing_sawyer', 'scratch_itch_jaco', 'arm_manipulation_pr2', 'feeding_sawyer', 'bed_settle', 'drinking_jaco'}


def _names():
    from assistive_gym_amd.model.compiler import COMPILERS
    return [n if n in LEAN else pytest.param(n, marks=full) for n in sorted(COMPILERS)]


@pytest.mark.skipif(not os.path.isdir(ASSETS), reason='reference assets not on this box')
@pytest.mark.parametrize('name', _names())
def test_blob_on_disk_matches_the_compiler(name):
    import json
    from assistive_gym_amd.blob import DATA_DIR
    from assistive_gym_amd.model.compiler import COMPILERS, VERSION
    words, meta = COMPILERS[name]()
    disk = np.fromfile(os.path.join(DATA_DIR, name + '.agxblob'), dtype=np.uint32)
    assert int(disk[1]) == VERSION
    assert len(disk) == len(words) and np.array_equal(disk, np.asarray(words, dtype=np.uint32)), 'recompile: python -m assistive_gym_amd.model.compiler ' + name
    on_disk = json.load(open(os.path.join(DATA_DIR, name + '.meta.json')))
    assert on_disk['header'] == json.loads(json.dumps(meta['header']))


def test_every_env_id_has_its_blob():
    from assistive_gym_amd.blob import DATA_DIR
    from assistive_gym_amd.envs import ENV_IDS
    from assistive_gym_amd.model.compiler import COMPILERS
    for env_id, cls in ENV_IDS.items():
        assert cls.model in COMPILERS and os.path.exists(os.path.join(DATA_DIR, cls.model + '.agxblob')), env_id
    assert len(ENV_IDS) == 70          # the reference's 6 tasks x 6 robots x {-, Human} = 72, minus ArmManipulationStretch(Human), which raises in the reference's first step
