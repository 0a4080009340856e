"""Drop-in `assistive_gym` package backed by the MI355X stepper (assistive_gym_amd).

Put `<repo>/assistive_gym_amd/shim` on PYTHONPATH (or call `assistive_gym_amd.shim.install()`) and the reference's own
entry points keep working unchanged: `gym.make('assistive_gym:FeedingJaco-v1')`, `importlib.import_module('assistive_gym.envs')`
+ `getattr(module, '<Task><Robot>HumanEnv')` (assistive_gym/learn.py:61-69), `python -m assistive_gym.env_viewer`-style loops.
Only the environments whose hot path is built are registered (assistive_gym_amd.envs.ENV_IDS); asking for another id fails
the way gym fails for an unknown id.
"""
from assistive_gym_amd import envs as _envs

MAX_EPISODE_STEPS = 200          # the reference registers every id with max_episode_steps=200


def _register_all():
    try:
        from gym.envs.registration import register
    except Exception:            # gym is optional: assistive_gym_amd.envs.make() covers the same ids
        register = None
    try:
        from ray.tune.registry import register_env
    except Exception:
        register_env = None
    for env_id, cls in _envs.ENV_IDS.items():
        if cls.coop:
            # co-op envs are RLlib multi-agent envs, registered under 'assistive_gym:<id>' (e.g. feeding_envs.py:67)
            if register_env is not None:
                register_env('assistive_gym:' + env_id, lambda config, _c=cls: _c())
        elif register is not None:
            try:
                register(id=env_id, entry_point='assistive_gym.envs:%s' % cls.__name__, max_episode_steps=MAX_EPISODE_STEPS)
            except Exception:    # already registered (module re-import)
                pass


_register_all()
