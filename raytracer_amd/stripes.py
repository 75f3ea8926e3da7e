"""Row-stripe partition of a frame over ranks, and the gather that reassembles it.

The frame is cut into stripes of `stripe_rows` image rows; stripe s belongs to rank s % world (the same
rule rt_render applies to an rt_region).  Pixels are independent in the reference's render loop
(src/render.cpp:27-44), so the ranks exchange nothing while rendering; the only exchange is this
all-gather of finished stripes (RCCL on GPUs, gloo in the CPU tests).  Bloom needs the whole frame
(its filter reaches 2F rows up and down, src/extra.cpp:330-416) and runs after the gather.

Frames are in Screen order: memory row m holds image row H-1-m (src/screen.cpp:119-122), so image
stripe s is memory row block n_stripes-1-s when H is a multiple of stripe_rows.

Two exchanges: `gather` leaves the whole frame on every rank (one all-gather); `assemble` takes the K frames of
one launch sequence (rt_render_frames) and leaves frame f whole on rank f % world only (one all-to-all: every rank
sends each stripe once instead of world - 1 times), so that bloom and the copy to the host happen once per frame.
"""
import torch


class StripePlan:
    def __init__(self, height: int, width: int, rank: int, world: int, stripe_rows: int = 8, channels: int = 3,
                 device="cpu", dtype=torch.float32):
        if height % stripe_rows:
            raise ValueError(f"height {height} must be a multiple of the {stripe_rows}-row stripe")
        if not (0 <= rank < world):
            raise ValueError(f"rank {rank} outside world of {world}")
        self.rank, self.world = rank, world
        self.n_stripes = height // stripe_rows
        self.per_rank = (self.n_stripes + world - 1) // world
        self.block = stripe_rows * width * channels
        self.mine = torch.arange(rank, self.n_stripes, world, device=device)            # image stripes of this rank
        self.my_blocks = self.n_stripes - 1 - self.mine                                 # their memory row blocks
        # recv[r, k] is stripe k*world + r; padded slots (beyond n_stripes) are dropped after the gather
        order = torch.arange(self.per_rank * world, device=device)
        self.all_blocks = (self.n_stripes - 1 - order.clamp(max=self.n_stripes - 1))[: self.n_stripes]
        self.send = torch.zeros((self.per_rank, self.block), dtype=dtype, device=device)
        self.recv = torch.zeros((world, self.per_rank, self.block), dtype=dtype, device=device)

    def owner(self, image_row: int, stripe_rows: int = 8) -> int:
        return (image_row // stripe_rows) % self.world

    def gather(self, frame: torch.Tensor, dist) -> None:
        """frame: (H, W, C) Screen-order tensor holding this rank's stripes; on return it holds all of them."""
        if self.world == 1:
            return
        blocks = frame.view(self.n_stripes, self.block)
        self.send[: len(self.mine)] = blocks[self.my_blocks]
        dist.all_gather_into_tensor(self.recv.view(-1), self.send.view(-1))
        blocks[self.all_blocks] = self.recv.permute(1, 0, 2).reshape(self.per_rank * self.world, -1)[: self.n_stripes]

    def root_of(self, frame_index: int) -> int:
        """The rank on which `assemble` leaves frame `frame_index` of a sequence."""
        return frame_index % self.world

    def assemble(self, frames: torch.Tensor, dist) -> list:
        """frames: (K, H, W, C) Screen-order tensor holding this rank's stripes of K frames.  On return the frames
        f with f % world == rank hold every rank's stripes; the others are unchanged.  Returns those f."""
        K = frames.shape[0]
        mine_f = list(range(self.rank, K, self.world))
        if self.world == 1:
            return mine_f
        per_root = (K + self.world - 1) // self.world                    # frames a rank is the root of (padded)
        key = (K, frames.device, frames.dtype)
        if getattr(self, "_a2a_key", None) != key:
            shape = (per_root, self.world, self.per_rank, self.block)    # [j, destination]: frame j * world + destination
            self._a2a_stage = torch.zeros(shape, dtype=frames.dtype, device=frames.device)
            self._a2a_send = torch.zeros((self.world, per_root, self.per_rank, self.block), dtype=frames.dtype, device=frames.device)
            self._a2a_recv = torch.zeros_like(self._a2a_send)                 # [source, j]
            self._a2a_key = key
        blocks = frames.view(K, self.n_stripes, self.block)
        self._a2a_stage.view(per_root * self.world, self.per_rank, self.block)[:K, : len(self.mine)] = blocks[:, self.my_blocks]
        self._a2a_send.copy_(self._a2a_stage.permute(1, 0, 2, 3))
        dist.all_to_all_single(self._a2a_recv.view(-1), self._a2a_send.view(-1))
        for j, f in enumerate(mine_f):
            # recv[s, j, p] is stripe p * world + s of frame f
            blocks[f][self.all_blocks] = self._a2a_recv[:, j].permute(1, 0, 2).reshape(self.per_rank * self.world, -1)[: self.n_stripes]
        return mine_f
