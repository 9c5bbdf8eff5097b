from videoswap_amd.utils import save_images_as_gif, save_video_to_dir  # noqa: F401
