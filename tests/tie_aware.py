"""The tie-aware index comparison lives in oracle/tie_aware.py (checker code shared with __graft_entry__.smoke())."""
from oracle.tie_aware import *  # noqa: F401,F403
from oracle.tie_aware import _crop_la, _pix  # noqa: F401
