"""Host-side mirror of the reference's utils/ seams for the hot path (same function names, argument
meaning and return conventions), running on libcorsair_hip.so."""
