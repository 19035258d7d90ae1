"""The dist entry points import ConfigParser from this module name
(/root/reference/OATrans/train_dist_multi.py:8); one implementation serves both."""
try:
    from OATrans.parse_config import ConfigParser  # noqa: F401
except ImportError:
    from parse_config import ConfigParser  # noqa: F401
