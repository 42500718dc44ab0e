"""MI355X-native dense RGB-D tracking core (drop-in for dvo_core's DenseTracker::match path)."""
__all__ = ["synth"]
