"""Algorithmic bytes of one frame: the one definition bench.py and the counter tools share."""

# candidates per method: the classic / direct8 / geodesic / bilateral-grid ranges are inclusive (numD + 1 planes)
INCLUSIVE_ALGORITHMS = (2, 3, 4, 5)


def candidates(algorithm, num_disparity):
    return num_disparity + 1 if int(algorithm) in INCLUSIVE_ALGORITHMS else num_disparity


def algorithmic_bytes(width, height, n_candidates):
    """Inputs read once (two 8UC3 images) + aggregated f32 cost volume written once + f32 disparity written once."""
    return 2 * width * height * 3 + width * height * n_candidates * 4 + width * height * 4
