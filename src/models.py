from pulpo_amd.models import *  # noqa: F401,F403
from pulpo_amd.models import PULPo  # noqa: F401
