"""Drop-in for the reference's ``skeletal_network/`` scripts (same module and symbol names)."""
