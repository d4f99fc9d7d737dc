"""Drop-in shim: with ``factor-graph-neural-network_amd/`` first on PYTHONPATH the reference's
``from lib.model.mpnn import factor_mpnn, FactorNN`` (train_ldpc.py:13) resolves to the
MI355X-native classes.  The reference's ``lib.data`` (dataset generation) is out of scope."""
