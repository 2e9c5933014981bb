"""MI355X-native VPDQ hashing + pairwise similarity (drop-in for `hvdaccelerators.vpdq`
and the reference's hashing facade). Import as `hvd_amd` (see /hvd_amd.py)."""

__version__ = "0.1.0"
