from echoseal_amd.fastpolar import *  # noqa: F401,F403  (drop-in alias of the reference's module name)
from echoseal_amd import fastpolar as _impl
globals().update({k: v for k, v in vars(_impl).items() if not k.startswith('__')})
