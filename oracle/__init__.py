"""oracle/: the CPU checker (TEST INFRASTRUCTURE ONLY).  check.py = Python handle on liboracle.so; ref_shim/ + Makefile.ref = the
reference's own code compiled in place (oracle/_ref/libref.so), which pins the oracle to the reference."""
