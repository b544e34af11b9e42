raise ImportError("tests/no_torch: torch is unimportable in this process on purpose (bench.py's launcher branch must not need it)")
