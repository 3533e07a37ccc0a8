"""ORACLE — test infrastructure only (see oracle/README.md).  Importable from tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg; never from stereospike_amd/."""
