"""Hand-computed known-answer test for DENSE (non-hashed) 3-D levels of the temporal grid: every number below follows
from the reference kernel's text (NS/field_components/cuda/csrc/temporal_gridencoder.cu:62-88 get_grid_index, :107-203 kernel_grid;
temporal_grid.py:211-228 level sizes, :320-330 get_temporal_index) by pencil, not from the oracle or the HIP kernel -- both are checked AGAINST it
(tests/test_oracle_golden.py on the CPU, tests/test_gpu_tgrid.py on the GPU).

Encoder: input_dim 3, num_levels 2, level_dim 1, temporal_dim 3, base_resolution 2, per_level_scale 2, log2_hashmap_size 19, align_corners False.
  level 0: scale = 2^0 * 2 - 1 = 1, resolution = ceil(1) + 1 = 2, rows = (2 + 1)^3 = 27 -> 32 (multiple of 8); strides 1, 3, 9 -> the stride after
           the loop is 27 <= 32: NO hash for either gridtype; row = x + 3 y + 9 z.
  level 1: scale = 2^1 * 2 - 1 = 3, resolution = 4, rows = (4 + 1)^3 = 125 -> 128 at offset 32; strides 1, 5, 25; row = 32 + x + 5 y + 25 z.
  embedding columns: level_dim + temporal_dim = 4; the channel table has temporal_dim - 1 = 2 rows.
Point p = (0.25, 0.5, 0.75):
  level 0: pos = p * 1 + 0.5 = (0.75, 1.0, 1.25): cell (0, 1, 1), fractions (0.75, 0, 0.25).  Corners with non-zero weight
           (x, y, z) -> row, weight:  (0,1,1) -> 12, 0.25 * 1 * 0.75 = 0.1875;  (1,1,1) -> 13, 0.5625;  (0,1,2) -> 21, 0.0625;  (1,1,2) -> 22, 0.1875;
           the y + 1 corners (rows 15, 16, 24, 25) carry weight 0.
  level 1: pos = p * 3 + 0.5 = (1.25, 2.0, 2.75): cell (1, 2, 2), fractions (0.25, 0, 0.75): rows 32 + {61, 62, 86, 87}, weights 0.1875, 0.0625, 0.5625, 0.1875.
Time: t = 0 -> table row 0, weight_a = 1: the channel reads column 0 only.  t = 0.25 -> v = 0.25 * (2 - 1), table row 0: 0.75 * column A + 0.25 *
  column B with A = 0 (nothing has replaced channel 0 yet) and B = level_dim + 0 = 1 (the column entering in row 0).
Embedding: column 0 = (local row)^2, column 1 = 100 + local row, other columns 7 (never read at these times).
  t = 0:    level 0 = 0.1875 * 144 + 0.5625 * 169 + 0.0625 * 441 + 0.1875 * 484 = 240.375
            level 1 = 0.1875 * 3721 + 0.0625 * 3844 + 0.5625 * 7396 + 0.1875 * 7569 = 6517.375
  t = 0.25: column 1 is linear in the row, so its trilinear value is 100 + the interpolated row: level 0: 100 + 0.75 + 3 * 1 + 9 * 1.25 = 115;
            level 1: 100 + 1.25 + 5 * 2 + 25 * 2.75 = 180;  outputs 0.75 * 240.375 + 0.25 * 115 = 209.03125 and 0.75 * 6517.375 + 0.25 * 180 = 4933.03125.
Out of range (any coordinate outside [0, 1]): both outputs 0.
Gradient of sum(outputs) at p, t = 0.25 w.r.t. the embedding: rows above get (0.75 w, 0.25 w) in columns (0, 1); everything else exactly 0."""
KW = dict(temporal_dim=3, input_dim=3, num_levels=2, level_dim=1, per_level_scale=2.0, base_resolution=2, log2_hashmap_size=19, desired_resolution=None,
          align_corners=False)
OFFSETS = [0, 32, 160]
POINT = (0.25, 0.5, 0.75)
OUT_T0 = (240.375, 6517.375)
OUT_T025 = (209.03125, 4933.03125)
CORNERS = {12: 0.1875, 13: 0.5625, 21: 0.0625, 22: 0.1875, 32 + 61: 0.1875, 32 + 62: 0.0625, 32 + 86: 0.5625, 32 + 87: 0.1875}


def embedding(torch):
    emb = torch.full((160, 4), 7.0)
    local = torch.cat([torch.arange(32), torch.arange(128)]).float()
    emb[:, 0] = local * local
    emb[:, 1] = 100.0 + local
    return emb


def inputs(torch):
    x = torch.tensor([POINT, POINT, (0.25, 1.5, 0.75), (-0.01, 0.5, 0.5)], dtype=torch.float32)
    t = torch.tensor([[0.0], [0.25], [0.25], [0.0]], dtype=torch.float32)
    want = torch.tensor([OUT_T0, OUT_T025, (0.0, 0.0), (0.0, 0.0)], dtype=torch.float32)
    return x, t, want


def expected_grad(torch):
    g = torch.zeros(160, 4)
    for row, w in CORNERS.items():
        g[row, 0], g[row, 1] = 0.75 * w, 0.25 * w
    return g
