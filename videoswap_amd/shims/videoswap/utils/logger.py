from videoswap_amd.utils import dict2str, get_time_str, set_path_logger  # noqa: F401
