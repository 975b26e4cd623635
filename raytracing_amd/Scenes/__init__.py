"""Scene definitions: the reference's four example scenes plus the two
synthetic benchmark scenes (spotlight easing scene, many-sphere scene)."""
