"""Loading of the committed golden fixtures: see dqmc_amd/fixtures.py."""
from dqmc_amd.fixtures import GOLD, NAMES, g0_error, load  # noqa: F401
