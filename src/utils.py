from pulpo_amd.utils import ModuleIntDict  # noqa: F401
