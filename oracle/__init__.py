"""ORACLE - test infrastructure only (see oracle/cpu_ref.py header)."""
