from videoswap_amd.utils import MessageLogger, dict2str, get_time_str, reduce_loss_dict, set_path_logger  # noqa: F401
