"""Oracle: CPU restatement of the reference path. Test infrastructure only (see restatement.py)."""
