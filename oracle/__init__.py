"""CPU oracle — test infrastructure only (see oracle/README.md)."""
