"""Test-side alias of phyml_amd.phyg (the PHYG named-array container reader/writer)."""
from phyml_amd.phyg import load, save  # noqa: F401
