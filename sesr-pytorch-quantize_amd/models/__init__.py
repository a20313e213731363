"""Network topologies of the integer path (collapsible-conv SESR / NRDM nets)."""
