"""Score-network front end: the engine-backed NCSN++ and the reference's model helper API."""
