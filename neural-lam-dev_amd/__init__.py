"""MI355X-native InteractionNet / GraphLAM / Hi-LAM hot path."""
