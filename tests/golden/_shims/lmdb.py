"""Stand-in for py-lmdb (not installed here): lav_amd.data.lmdb_ro, this repository's read-only reader, behind py-lmdb's
`open(...).begin(write=False).get(key)` surface - what lav/utils/datasets/basic_dataset.py:47-55 calls."""
from lav_amd.data.lmdb_ro import Environment, Error, Transaction, open  # noqa: F401,A004
