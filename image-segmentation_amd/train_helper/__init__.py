"""Training harness around the MI355X segmentation step: the loop either side of the hot path."""
