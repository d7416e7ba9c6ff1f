"""Host-side helpers around the rasterizer path: camera conventions and synthetic scenes."""
from .cameras import MiniCam, camera_from_RT, look_at_camera, fibonacci_cameras, fov2focal, focal2fov, \
    world_to_view, projection_matrix
from .synthetic import RawGaussians, make_gaussians, make_config, CONFIGS
from .model import GaussianModel
from .sh import eval_sh, RGB2SH, SH2RGB
from .losses import l1_loss, psnr, training_loss_fused
from .parallel import init_from_env, shard_views, GradBucket, ShardedStep, reduce_densification_stats, \
    rank1_sh_exchange, exchange_bytes_per_gaussian
from .trainer import Trainer
from .io import save_ply, load_ply, read_ply_vertices, capture, restore
