"""offmark-compatible package of the MI355X frame-watermark engine."""
