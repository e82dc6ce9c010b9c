"""Device-side pieces of the reference's tasks/ that sit between detection and the metrics."""
