from echoseal_amd.polar_fast import *  # noqa: F401,F403  (drop-in alias of the reference's module name)
from echoseal_amd import polar_fast as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
